"""Reynolds flocking expert: the policy the reference's data generator drives the env with
(examples/simulating_data/helper/Reynolds.py, helper/Reynolds_Node.py, gen_data.py:33), as one fused HIP
kernel (mrs_reynolds) over E envs.

`Reynolds(N, D, K=1, OUT_DIM=3)` keeps the reference constructor and `forward(As, Xs)` = `forward_batch` with
ITS input layout `Xs: (batch, N, D, K+1)` (Reynolds.py:80-82; that is the README's layout -- MRS.step itself
returns (K+1, N, D), MRS.py:99).  `forward_stack(Xk)` takes this library's stack `(E, K+1, N, D)` instead, and
`from_env(env)` reads slot 1 of the env's history ring in place (no copy).  As in the reference, the adjacency
argument is ignored (forward_batch substitutes ones - eye, Reynolds.py:83) and the states of the PREVIOUS step
are used (:87).  Any K >= 1: forward_batch aggregates hops 2..K (Reynolds.py:89-97), but its controller reads hops
0 and 1 only (`Z[..., 1] - Z[..., 0]`, Reynolds_Node.py:30) and of those the first six state components, so the
action is a function of X(t-1) alone whatever K is (tests/golden/F7 holds K = 2 and K = 3 outputs of the reference).
"""
import torch

from . import native


class Reynolds:
    def __init__(self, N, D, K=1, OUT_DIM=3, device="cuda"):
        if K < 1 or OUT_DIM != 3:
            raise NotImplementedError("Reynolds: K >= 1 and OUT_DIM = 3 (Reynolds_Node.py:26-38 returns 3-vectors)")
        if D < 6:
            raise ValueError("Reynolds: D must be >= 6 (rel_state_size, Reynolds.py:15)")
        self.N, self.D, self.K, self.OUT_DIM = int(N), int(D), int(K), int(OUT_DIM)
        self.rel_state_size = 6
        self.device = torch.device(device)
        if self.device.type == "cuda" and self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device() if torch.cuda.is_available() else 0)
        self._shards = {}
        self.forward = self.forward_batch

    def _shard(self, E):
        sh = self._shards.get(E)
        if sh is None:
            sh = self._shards[E] = native.SwarmShard(E, self.N, self.device)
        return sh

    def __call__(self, As, Xs):
        return self.forward_batch(As, Xs)

    def forward_batch(self, As, Xs):  # Reynolds.py:80-106
        Xs = torch.as_tensor(Xs)
        if Xs.dim() != 4 or Xs.shape[1] != self.N or Xs.shape[3] != self.K + 1:
            raise ValueError("Xs must be (batch, N=%d, D, K+1=%d), got %s" % (self.N, self.K + 1, tuple(Xs.shape)))
        x_prev = Xs[:, :, :, 1].to(device=self.device, dtype=torch.float32).contiguous()
        return self._shard(x_prev.shape[0]).reynolds(x_prev)

    def forward_stack(self, Xk):
        """Xk: this library's observation stack (E, K+1, N, D) (or (K+1, N, D) for one env)."""
        Xk = torch.as_tensor(Xk)
        if Xk.dim() == 3:
            Xk = Xk.unsqueeze(0)
        if Xk.shape[1] < 2:
            raise ValueError("forward_stack needs at least the slices t and t-1 (K_HOPS >= 1)")
        x_prev = Xk[:, 1].to(device=self.device, dtype=torch.float32).contiguous()
        return self._shard(x_prev.shape[0]).reynolds(x_prev)

    def from_env(self, env, out=None):
        """Expert action for `env`'s current history (an mrsgym_amd.MRS with K_HOPS >= 1): reads the previous
        step's slice straight from the history ring and uses the env's own native handle."""
        if env.K_HOPS < 1:
            raise ValueError("Reynolds needs K_HOPS >= 1 (it acts on X(t-1))")
        ring = env._Xring
        a = env.shard.reynolds(ring.buf[ring.head + 1], out)
        return a if env.N_ENVS > 1 else a[0]
