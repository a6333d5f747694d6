"""Host-side mirror of the reference's Gym surface: class MRS ('mrs-v0'), make().

Same names, kwargs, return conventions and error behaviour as mrsgym/MRS.py (cited per method), with
every per-agent Python loop of the reference replaced by one C-ABI call into the HIP library.
Extensions (all optional, defaults reproduce the reference's shapes): N_ENVS, DEVICE, OBS,
A_FORMAT, SEED, RESET_CONTROLLERS, CHECK_NAN, ENV_INDEX_BASE.  With N_ENVS = E > 1 every tensor
gains a leading E axis: Xk (E,K+1,N,D), Ak (E,K+1,N,N), actions (E,N,ACTION_DIM).
"""
import math
import time
import types
import warnings

import numpy as np
import torch

from . import native
from .facade import Environment, QuadView, StateFnCompiler
from .history import HistoryRing
from .lazy import LazyDenseA

try:  # gym / gymnasium are optional (absent in the build image)
    import gym as _gym  # type: ignore
    from gym.spaces import Box  # type: ignore
    _EnvBase = _gym.Env
except Exception:  # pragma: no cover
    try:
        import gymnasium as _gym  # type: ignore
        from gymnasium.spaces import Box  # type: ignore
        _EnvBase = _gym.Env
    except Exception:
        _gym = None
        _EnvBase = object

        class Box:  # minimal stand-in: low/high/shape/dtype, sample()
            def __init__(self, low, high, dtype=np.float32):
                self.low, self.high = np.asarray(low), np.asarray(high)
                self.shape, self.dtype = self.low.shape, dtype

            def sample(self):
                lo = np.where(np.isfinite(self.low), self.low, -1.0)
                hi = np.where(np.isfinite(self.high), self.high, 1.0)
                return np.random.uniform(lo, hi).astype(self.dtype)

            def contains(self, x):
                x = np.asarray(x)
                return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))


class MRS(_EnvBase):

    metadata = {'render.modes': ['headless', 'bullet']}

    # MRS.__init__, MRS.py:21-66
    def __init__(self, state_fn=None, reward_fn=None, done_fn=None, info_fn=None, update_fn=None, start_fn=None,
                 env='simple', **kwargs):
        if _EnvBase is not object:
            super(MRS, self).__init__()
        # Constants (MRS.py:24-34)
        self.N_AGENTS = 1
        self.K_HOPS = 0
        self.STATE_SIZE = 0
        self.ACTION_DIM = 0
        self.AGENT_RADIUS = 0.3
        self.COMM_RANGE = float('inf')
        self.RETURN_A = None          # reference: declared, never read (A is always returned). None = that.
        self.RETURN_EVENTS = False
        self.ACTION_TYPE = "set_target_vel"
        self.HEADLESS = False
        self.MAX_TIMESTEPS = float('inf')
        # extensions
        self.N_ENVS = 1
        self.DEVICE = "cuda"
        self.OBS = None               # explicit fused observation spec, e.g. ("pos", "vel")
        self.A_FORMAT = "dense"       # "dense": float32 0/1 (reference) | "packed": int64 bit rows (E,K+1,N,ceil(N/64))
        self.SEED = 0
        self.ENV_INDEX_BASE = 0       # global index of this shard's first env (multi-GPU)
        self.RESET_CONTROLLERS = False  # reference: PID integrators survive reset() (QuadControl objects persist)
        self.CHECK_NAN = None         # "sync" | "lazy" | "off"; None = sync for N_ENVS==1 else lazy
        self.ROUND_EULER_READBACK = False   # True: the controller's literal float32 rounding of the Euler angles (include/mrs_hip.h)
        self.QUAD_CONTACT = True      # quad-quad contact (sphere model, MrsParams.pair_contact)
        self.SOLVER_ITERS = None      # cap of the ground-contact sweeps per body and step (None = the library's default, 10)
        self.REST_SHORTCUT = True     # bodies lying flat at rest finished in their own lane (MrsParams.rest_shortcut)
        self.HISTORY_SLOTS = 0        # ring length; 0 = 16*(K_HOPS+1)
        self.COPY_OUTPUTS = None      # None = clone returned stacks iff N_ENVS == 1 (reference returns fresh tensors)
        self.AUTO_RESET = False       # vectorised loops: envs whose `done` is set are reset inside step() (reset_envs)
        # BulletSim constants (BulletSim.py:11-15); DT/GRAVITY reach the world only, never the controller
        self.REAL_TIME = False
        self.GRAVITY = 9.81
        self.DT = 0.01
        self.set_constants(kwargs)
        # Inputs (MRS.py:37-42)
        self.state_fn = state_fn
        self.reward_fn = reward_fn if (reward_fn is not None) else (lambda **kwargs: 0.0)
        self.done_fn = done_fn if (done_fn is not None) else (lambda **kwargs: kwargs["steps_since_reset"] >= self.MAX_TIMESTEPS)
        self.info_fn = info_fn if (info_fn is not None) else (lambda **kwargs: {})
        # the defaults by identity: step() inlines them (their results need no kwargs dict and no three Python calls -- a
        # microsecond of the host's ~8 per step, which is what a small swarm's step waits for); a callback assigned later is seen
        self._default_cbs = (self.reward_fn if reward_fn is None else None, self.done_fn if done_fn is None else None,
                             self.info_fn if info_fn is None else None)
        self.update_fn = update_fn
        self.start_fn = start_fn
        if not isinstance(env, str):
            raise NotImplementedError("only env='simple' (N quadcopters over the ground box, EnvCreator.py:7-13) is "
                                      "on the accelerated path; custom Environment objects are out of scope")
        if env != 'simple':
            raise ValueError("unknown envtype %r" % (env,))
        self.device = torch.device(self.DEVICE)
        if self.device.type == "cuda" and self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device() if torch.cuda.is_available() else 0)
        # sim handle (BulletSim.__init__/setup) + env_generator('simple')
        params = native.default_params()
        params.gravity, params.dt = float(self.GRAVITY), float(self.DT)
        params.round_euler_readback = int(bool(self.ROUND_EULER_READBACK))
        params.pair_contact = int(bool(self.QUAD_CONTACT))
        params.rest_shortcut = int(bool(self.REST_SHORTCUT))
        if self.SOLVER_ITERS is not None:
            if int(self.SOLVER_ITERS) < 1:
                raise ValueError("SOLVER_ITERS must be >= 1")
            params.solver_iters = int(self.SOLVER_ITERS)
        self.sim = types.SimpleNamespace(DT=float(self.DT), GRAVITY=float(self.GRAVITY), REAL_TIME=bool(self.REAL_TIME),
                                         HEADLESS=bool(self.HEADLESS), id=0, params=params)
        self.shard = native.SwarmShard(self.N_ENVS, self.N_AGENTS, self.device, params=params)
        self.env = Environment(self)
        # Constants that depend on other constants (MRS.py:51-57; the size expression keeps upstream's precedence)
        n_obs = self.K_HOPS + 1 * self.N_AGENTS * self.STATE_SIZE
        self.observation_space = Box(np.full((n_obs), -np.inf, dtype=np.float32), np.full((n_obs), np.inf, dtype=np.float32))
        self.action_space = Box(np.tile(np.array([9.81 - 1, -1., -1., -1.]), self.N_AGENTS),
                                np.tile(np.array([9.81 + 1, 1., 1., 1.]), self.N_AGENTS))
        self.START_POS = None         # None = default_spawn_dist() semantics on the device (MRS.py:69-78)
        self.START_ORI = torch.tensor([0, 0, -np.pi / 2, 0, 0, np.pi / 2])
        if len(self.START_ORI.shape) == 1:
            self.START_ORI = self.START_ORI.expand(self.N_AGENTS, -1)
        self.set_constants(kwargs)
        # observation plumbing
        self._obs = StateFnCompiler(self)
        self.STATE_DIM = None
        # Data (MRS.py:59-65)
        self.steps_since_reset = 0
        self.last_action = None
        self.last_obs = None
        self.last_loop_time = time.monotonic()
        self.is_initialised = False
        self._resets = 0
        self._global_step = 0
        self._reset_step = torch.zeros(self.N_ENVS, dtype=torch.int64, device=self.device)  # per env: _global_step at its last reset
        self._alloc_history()
        self.reset()

    # ------------------------------------------------------------------ helpers
    def set_constants(self, kwargs):  # MRS.py:81-84
        for name, val in kwargs.items():
            if name in self.__dict__:
                self.__dict__[name] = val

    @property
    def E(self):
        return self.N_ENVS

    def _squeeze(self, t):
        """(E, ...) -> (...) when N_ENVS == 1 so that shapes equal the reference's."""
        return t[0] if self.N_ENVS == 1 else t

    def _stack_view(self, w):
        """ring window (K+1, E, ...) -> the reference's stack layout (E, K+1, ...) / (K+1, ...) for one env."""
        return self._squeeze(w.permute(1, 0, 2, 3))

    def _copies(self):
        return self.COPY_OUTPUTS if self.COPY_OUTPUTS is not None else (self.N_ENVS == 1)

    def _out(self, t):
        # the reference returns fresh tensors every step; views into the history ring are overwritten
        # HISTORY_SLOTS steps later (or at the next reset), which is what a vectorised loop wants
        copy = self.COPY_OUTPUTS if self.COPY_OUTPUTS is not None else (self.N_ENVS == 1)
        return t.clone() if copy else t

    def _alloc_history(self):
        E, N, W = self.N_ENVS, self.N_AGENTS, self.shard.W
        self._Xring = None            # allocated once D is known (StateFnCompiler)
        self._Apacked = HistoryRing(self.K_HOPS, (E, N, W), torch.int64, self.device, self.HISTORY_SLOTS, pad="zero",
                                    view_fn=self._stack_view)
        # A_FORMAT = "dense": the float32 ring exists (and is written by the step kernel) only while info["A"] is being READ
        # (lazy.LazyDenseA, _materialize_A): allocated at the first read, dropped back to packed-only after DENSE_IDLE_STEPS
        # steps without one
        self._Adense = None
        self._dense_live = False
        self._dense_read_step = -(1 << 60)
        self._a_stamp = 0             # counts the changes of the adjacency window: a LazyDenseA knows whether it is still the current one

    def _ensure_xbuf(self, D):
        if self._Xring is None or self._Xring.buf.shape[-1] != D:
            self._Xring = HistoryRing(self.K_HOPS, (self.N_ENVS, self.N_AGENTS, D), torch.float32, self.device,
                                      self.HISTORY_SLOTS, pad="copy", view_fn=self._stack_view)
            self.STATE_DIM = D

    # ------------------------------------------------------------------ observation / adjacency stacks
    def calc_Xk(self):  # MRS.py:87-95
        if not self._obs._compiled:
            self._obs.compile()
        if self._Xring is None:      # generic state_fn: D is only known after one evaluation
            X = self._obs.evaluate(self.state_fn)
            self._ensure_xbuf(X.shape[-1])
        self._obs.write_into(self._Xring.buf[self._Xring.next_slot()])
        self._Xring.committed()
        return self.get_Xk()

    def get_Xk(self):  # MRS.py:98-99  Xk: K+1 x N x D  (E x K+1 x N x D)
        return self._out(self._Xring.view())

    DENSE_IDLE_STEPS = 64

    def calc_Ak(self):  # MRS.py:102-110
        slot = self._Apacked.next_slot()
        dr = self._Adense if self._dense_live else None
        dslot = dr.next_slot() if dr is not None else 0
        self.shard.adjacency(self._Apacked.buf[slot], self.COMM_RANGE, dr.buf[dslot] if dr is not None else None)
        self._Apacked.committed()
        if dr is not None:
            dr.committed()
        self._a_stamp += 1
        return self.get_Ak()

    def _expand_newest_A(self):
        if self._dense_live:
            slot = self._Adense.next_slot()
            self.shard.adjacency_expand(self._Apacked.newest(), self._Adense.buf[slot])
            self._Adense.committed()

    def _activate_dense(self):
        """First read of info["A"] (or the first one after an idle stretch): the dense ring takes over the packed ring's position and
        its K+1 current slices -- one mrs_adjacency_expand over the window -- and the step kernel writes it from here on."""
        ar = self._Apacked
        if self._Adense is None:
            E, N = self.N_ENVS, self.N_AGENTS
            self._Adense = HistoryRing(self.K_HOPS, (E, N, N), torch.float32, self.device, ar.L, pad="zero", view_fn=self._stack_view)
        dr = self._Adense
        dr.head, dr.count = ar.head, ar.count
        self.shard.adjacency_expand(ar.window().contiguous(), dr.window().view(-1, self.N_AGENTS, self.N_AGENTS))
        self._dense_live = True

    def _materialize_A(self, stamp, saved):
        """The float32 stack behind a LazyDenseA made at window change number `stamp` (saved: a clone of the packed window it
        stood for, kept when outputs are copies; the window's position in the ring when they are views)."""
        if stamp == self._a_stamp:
            if not self._dense_live:
                self._activate_dense()
            self._dense_read_step = self._global_step
            return self._out(self._Adense.view())
        # an older stack, read late (the reference's loops do: DataGenerator.py logs the previous A after the step): from its own packed
        # rows -- the clone it kept (outputs are copies), or the ring slots it stood for as long as they have not been reused (views)
        if isinstance(saved, tuple):
            ar, (head, wraps) = self._Apacked, saved
            if not ar.intact(head, wraps):
                raise RuntimeError("info['A'] of an earlier step was first read after its history slots had been reused (outputs are views: "
                                   "N_ENVS > 1, COPY_OUTPUTS unset): read it within HISTORY_SLOTS - 2 (K_HOPS + 1) steps, or pass COPY_OUTPUTS=True")
            saved = ar.buf[head:head + ar.K + 1]
        K1, E, N = saved.shape[0], self.N_ENVS, self.N_AGENTS
        dense = torch.empty(K1, E, N, N, dtype=torch.float32, device=self.device)
        self.shard.adjacency_expand(saved, dense.view(-1, N, N))
        return self._stack_view(dense)

    def get_Ak(self):  # MRS.py:113-114  Ak: K+1 x N x N, missing slots are zeros (MRS.py:107-108)
        if self.A_FORMAT != "dense":
            return self._out(self._Apacked.view())
        stamp = self._a_stamp
        saved = self._Apacked.window().clone() if self._copies() else (self._Apacked.head, self._Apacked.wraps)
        E, K1, N = self.N_ENVS, self.K_HOPS + 1, self.N_AGENTS
        return LazyDenseA(lambda: self._materialize_A(stamp, saved), (K1, N, N) if E == 1 else (E, K1, N, N), self.device)

    def calc_A(self):  # MRS.py:117-124  newest adjacency only, no history side effect
        E, N, W = self.N_ENVS, self.N_AGENTS, self.shard.W
        packed = torch.zeros(E, N, W, dtype=torch.int64, device=self.device)
        if self.A_FORMAT != "dense":
            self.shard.adjacency(packed, self.COMM_RANGE)
            return self._squeeze(packed)
        dense = torch.zeros(E, N, N, dtype=torch.float32, device=self.device)
        self.shard.adjacency(packed, self.COMM_RANGE, dense)
        return self._squeeze(dense)

    def _clear_history(self):  # self.X = deque([]); self.A = deque([])  (MRS.py:185-186)
        for r in (self._Xring, self._Apacked, self._Adense if self._dense_live else None):
            if r is not None:
                r.clear()
        self._a_stamp += 1

    # ------------------------------------------------------------------ spawn
    def default_spawn_dist(self):  # MRS.py:69-78 (the torch.distributions object, for callers that want it)
        from torch.distributions import Normal, Uniform
        from .util import CombinedDistribution, SphereTransform
        from torch.distributions import TransformedDistribution
        z = Uniform(low=1.0 * torch.ones(self.N_AGENTS, 1), high=3.0 * torch.ones(self.N_AGENTS, 1))
        xy = TransformedDistribution(Normal(torch.zeros(self.N_AGENTS, 2), 1.0), [SphereTransform(radius=1.0, within=True)])
        return CombinedDistribution([xy, z], mixer='cat', dim=1)

    def get_relative_position(self, pos):  # MRS.py:166-170
        N = pos.shape[-2]
        return pos.unsqueeze(-2).expand(*pos.shape[:-2], N, N, 3) - pos.unsqueeze(-3).expand(*pos.shape[:-2], N, N, 3)

    @staticmethod
    def _draw(dist, k):
        """k independent samples of `dist`, stacked on a new leading axis -- one batched call when the distribution
        supports sample_shape (torch's own and util.CombinedDistribution do), k calls otherwise."""
        one = dist.sample()
        # torch sizes its CPU thread pool by the machine, not by the container's share of it: a ~1 M element normal_() on
        # 128 threads of a 16-core share stalls for ~90 ms every few calls (measured); a handful of threads take 2 ms
        nt = torch.get_num_threads()
        if nt > 8:
            torch.set_num_threads(4)
        try:
            try:
                x = dist.sample((k,))
                if tuple(x.shape) == (k,) + tuple(one.shape):
                    return x
            except Exception:
                pass
            return torch.stack([dist.sample() for _ in range(k)])
        finally:
            if nt > 8:
                torch.set_num_threads(nt)

    @staticmethod
    def _dist_to(d, dev):
        """The same distribution with its parameter tensors on `dev` (torch.distributions has no .to()): util.CombinedDistribution
        and TransformedDistribution recursively, any torch Distribution through its arg_constraints.  Raises for what it
        cannot rebuild; a transform that keeps tensors of its own fails at the first sample instead -- both are caught by
        _spawn_user, which then samples on the host as before."""
        from torch.distributions import Distribution, TransformedDistribution
        from .util import CombinedDistribution
        if isinstance(d, CombinedDistribution):
            return CombinedDistribution([MRS._dist_to(x, dev) for x in d.dist], mixer=d.mixer, dim=d.dim)
        if isinstance(d, TransformedDistribution):
            return TransformedDistribution(MRS._dist_to(d.base_dist, dev), list(d.transforms))
        if isinstance(d, Distribution) and d.arg_constraints:
            args = {k: getattr(d, k) for k in d.arg_constraints if k in d.__dict__}
            if not args:
                raise TypeError("no parameters found")
            return type(d)(**{k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in args.items()}, validate_args=False)
        raise TypeError("cannot move %r to the device" % (type(d),))

    def _start_pos_on_device(self):
        """(distribution on the device or None, samples are per agent) for the current START_POS, looked at once per object."""
        sp = self.START_POS
        c = getattr(self, "_sp_cache", None)
        if c is None or c[0] is not sp:
            per_agent = sp.sample().dim() == 1      # (3,) per-agent samples vs (N,3) joint samples (MRS.py:129-132)
            dev_dist = None
            if self.device.type == "cuda":
                try:
                    dev_dist = self._dist_to(sp, self.device)
                    x = dev_dist.sample((2,))
                    if x.device.type != "cuda" or tuple(x.shape[1:]) != tuple(sp.sample().shape):
                        dev_dist = None
                except Exception:
                    dev_dist = None
            c = self._sp_cache = (sp, dev_dist, per_agent)
        return c[1], c[2]

    def _spawn_user(self, env_mask=None, rounds=4):
        """MRS.generate_start_pos (MRS.py:127-154) for START_POS given as a distribution, for all (masked) envs at once:
        `rounds` re-sampling rounds per env are drawn up front and the greedy rejection runs on the device
        (mrs_spawn_from).  The samples come from a copy of the distribution whose parameters live on the device when one
        can be made (_dist_to: torch's own distributions, CombinedDistribution, TransformedDistribution) -- then for every
        env, selected or not, with no host round trip (a loop that resets a few envs every step, AUTO_RESET, spends 0.7
        instead of 7 ms per step that way: E = 512, N = 12, tools/probes/profile_autoreset.py) -- and from the caller's own
        object on the host otherwise, for the selected envs only.  Writes positions only."""
        sp, E, N, sh = self.START_POS, self.N_ENVS, self.N_AGENTS, self.shard
        dev_dist, per_agent = self._start_pos_on_device()
        mask = None if env_mask is None else torch.as_tensor(env_mask, device=self.device).to(torch.bool).reshape(E)
        sh.status.bitwise_and_(~(native.STATUS_SPAWN_MORE | native.STATUS_SPAWN_FAIL))
        resume = False
        for attempt in range(64 if dev_dist is not None else 0):
            x = dev_dist.sample((E * rounds * N,) if per_agent else (E * rounds,)).reshape(E, rounds, N, 3)
            sh.spawn_from(x.to(torch.float32), agent_radius=self.AGENT_RADIUS, env_mask=mask, resume=resume)
            more = (sh.status & native.STATUS_SPAWN_MORE) != 0      # one host sync per attempt
            if not bool(more.any()):
                return
            sh.status.bitwise_and_(~native.STATUS_SPAWN_MORE)
            mask, resume, rounds = more, True, min(64, rounds * 2)
        if dev_dist is not None:
            raise RuntimeError("START_POS: no layout with all pairs >= 2*AGENT_RADIUS = %.2f m apart after many re-sampling "
                               "rounds (the reference's generate_start_pos would still be looping, MRS.py:137-153)" % (2 * self.AGENT_RADIUS))
        for attempt in range(64):
            idx = torch.arange(E) if mask is None else torch.nonzero(mask.cpu()).flatten()
            if idx.numel() == 0:
                return
            n = int(idx.numel())
            x = self._draw(sp, n * rounds * N).reshape(n, rounds, N, 3) if per_agent else self._draw(sp, n * rounds).reshape(n, rounds, N, 3)
            cand = torch.zeros(E, rounds, N, 3, dtype=torch.float32) if n != E else None
            if cand is None:
                cand = x.to(torch.float32)
            else:
                cand[idx] = x.to(torch.float32)
            sh.spawn_from(cand.to(self.device), agent_radius=self.AGENT_RADIUS, env_mask=mask, resume=resume)
            more = (sh.status & native.STATUS_SPAWN_MORE) != 0      # one host sync per attempt
            if not bool(more.any()):
                return
            sh.status.bitwise_and_(~native.STATUS_SPAWN_MORE)
            mask, resume, rounds = more, True, min(64, rounds * 2)
        raise RuntimeError("START_POS: no layout with all pairs >= 2*AGENT_RADIUS = %.2f m apart after many re-sampling "
                           "rounds (the reference's generate_start_pos would still be looping, MRS.py:137-153)" % (2 * self.AGENT_RADIUS))

    def generate_start_pos(self):  # MRS.py:127-154
        """Start positions as a tensor: START_POS itself if it is one, else one batched draw + device-side rejection
        (the state buffers are used as scratch: reset()/reset_envs() call _spawn_user directly instead)."""
        E, N = self.N_ENVS, self.N_AGENTS
        sp = self.START_POS
        if isinstance(sp, torch.Tensor):
            return sp if sp.dim() == 3 else sp.unsqueeze(0).expand(E, N, 3)
        saved = self.shard.pos.clone()
        self._spawn_user()
        out = self.shard.view(self.shard.pos).to(torch.float32).clone()
        self.shard.pos.copy_(saved)
        return out

    def generate_start_ori(self):  # MRS.py:157-161
        so = torch.as_tensor(self.START_ORI)
        if so.dim() == 1:   # README.md:75-77: a (3) / (6) tensor applies to every agent
            so = so.expand(self.N_AGENTS, -1)
        if so.shape[-1] == 3:
            return so
        lo, hi = so[..., :3].to(torch.float32), so[..., 3:].to(torch.float32)
        shape = (self.N_ENVS,) + tuple(lo.shape) if self.N_ENVS > 1 else tuple(lo.shape)
        x = torch.rand(shape, device=self.device)
        return x * (hi - lo).to(self.device) + lo.to(self.device)

    def _default_spawn(self, env_mask=None):
        """START_POS=None: the reference's default_spawn_dist() + rejection, on the device (mrs_spawn)."""
        so = torch.as_tensor(self.START_ORI, dtype=torch.float32)
        if so.dim() == 2 and not bool((so == so[0]).all()):
            return False
        so = so.reshape(-1, so.shape[-1])[0]
        if so.shape[0] == 3:
            lo = hi = so.tolist()
        else:
            lo, hi = so[:3].tolist(), so[3:].tolist()
        self._resets += 1
        # only the spawn bit: a pending NaN-action flag must survive until check_errors() has raised it
        self.shard.status.bitwise_and_(~native.STATUS_SPAWN_FAIL)
        self.shard.spawn(seed=(int(self.SEED) << 20) + self._resets, env_index_base=self.ENV_INDEX_BASE,
                         agent_radius=self.AGENT_RADIUS, ori_lo=lo, ori_hi=hi, env_mask=env_mask)
        if int((self.shard.status & native.STATUS_SPAWN_FAIL).any()):
            raise RuntimeError("default spawn: %d agents of radius %.2f do not fit the unit-disc x [1,3] m volume "
                               "(the reference's generate_start_pos never terminates here, MRS.py:137-153); "
                               "pass START_POS" % (self.N_AGENTS, self.AGENT_RADIUS))
        return True

    # ------------------------------------------------------------------ Gym API
    def reset(self, pos=None, ori=None, vel=None, angvel=None):  # MRS.py:174-192
        if self.is_initialised:
            self.check_errors()      # a NaN action flagged since the last poll raises here, not never (MRS.py:247-248)
        self.is_initialised = True
        E, N = self.N_ENVS, self.N_AGENTS
        spawned = False
        if pos is None and self.START_POS is None:
            # default_spawn_dist() + rejection on the device; also draws START_ORI when it is one shared range
            spawned = self._default_spawn()
            if not spawned:
                saved, self.START_POS = self.START_POS, self.default_spawn_dist()
                self._spawn_user()
                self.START_POS = saved
        elif pos is None and not isinstance(self.START_POS, torch.Tensor):
            self._spawn_user()                       # positions are on the device already: set_state keeps them (pos=None)
        elif pos is None:
            pos = self.generate_start_pos()
        if ori is None and not spawned:
            ori = self.generate_start_ori()
            if ori.dim() == 2 and E > 1:
                ori = ori.unsqueeze(0).expand(E, N, ori.shape[-1])
        if vel is None:
            vel = torch.zeros(E, N, 3)
        if angvel is None:
            angvel = torch.zeros(E, N, 3)
        self.env.set_state(pos=pos, ori=ori, vel=vel, angvel=angvel)
        if self.RESET_CONTROLLERS:
            self.shard.pid_reset()
        self._clear_history()
        self.steps_since_reset = 0
        self._reset_step.fill_(self._global_step)
        if self.start_fn is not None:
            self.start_fn(self)
        Xk = self.calc_Xk()
        self.last_obs = Xk
        return Xk

    def reset_envs(self, env_mask, pos=None, ori=None, vel=None, angvel=None):
        """reset() for the envs selected by `env_mask` (bool, (E,)) only -- the vectorised counterpart of calling
        MRS.reset (MRS.py:174-192) on some of E independent reference instances.  The other envs keep their state,
        controller memory and K_HOPS history.  Selected envs: new start state (same START_POS / START_ORI rules),
        zero velocities, X history = K+1 copies of the new observation (MRS.py:92-93), A history = zeros
        (MRS.py:107-108; like reset(), no adjacency is computed until the next step).  State, controller memory and
        history are written through env masks on the device.  Returns the stacked Xk of all envs."""
        if not self.is_initialised:
            raise RuntimeError("reset_envs() before reset()")
        self.check_errors()
        E, N = self.N_ENVS, self.N_AGENTS
        mask = torch.as_tensor(env_mask, device=self.device).to(torch.bool).reshape(E)
        spawned = False
        if pos is None and self.START_POS is None:
            spawned = self._default_spawn(env_mask=mask)
            if not spawned:
                saved, self.START_POS = self.START_POS, self.default_spawn_dist()
                self._spawn_user(env_mask=mask)      # samples are drawn for the selected envs only
                self.START_POS = saved
        elif pos is None and not isinstance(self.START_POS, torch.Tensor):
            self._spawn_user(env_mask=mask)
        elif pos is None:
            pos = self.generate_start_pos()
        if ori is None and not spawned:
            ori = self.generate_start_ori()
            if ori.dim() == 2 and E > 1:
                ori = ori.unsqueeze(0).expand(E, N, ori.shape[-1])
        if not spawned:
            z3 = getattr(self, "_zeros3", None)      # on the device already: a host tensor would be built and copied per call
            if z3 is None:
                z3 = self._zeros3 = torch.zeros(E, N, 3, dtype=torch.float32, device=self.device)
            self.shard.set_state(pos=pos, ori=ori, vel=z3 if vel is None else vel, angvel=z3 if angvel is None else angvel, env_mask=mask)
        elif vel is not None or angvel is not None or ori is not None:
            self.shard.set_state(ori=ori, vel=vel, angvel=angvel, env_mask=mask)
        if self.RESET_CONTROLLERS:
            self.shard.pid_reset(env_mask=mask)
        self._reset_step[mask] = self._global_step
        xr = self._Xring
        self._obs.write_into(xr.buf[xr.head])          # newest slice in place: unchanged for the envs that were not reset
        if self.K_HOPS > 0:
            w = xr.window()
            w[1:, mask] = w[0, mask]
        for ring in (self._Apacked, self._Adense if self._dense_live else None):
            if ring is not None:
                ring.window()[:, mask] = 0
        self._a_stamp += 1
        Xk = self.get_Xk()
        self.last_obs = Xk
        return Xk

    def env_steps(self):
        """Steps since each env's own last reset, int64 (E,) on the device."""
        return self._global_step - self._reset_step

    def set(self, pos=None, ori=None, vel=None, angvel=None):  # MRS.py:196-205
        self.env.set_state(pos=pos, ori=ori, vel=vel, angvel=angvel)
        self._clear_history()
        self.steps_since_reset = 0
        if self.start_fn is not None:
            self.start_fn(self)
        Xk = self.calc_Xk()
        self.last_obs = Xk
        return Xk

    def set_data(self, name, val):  # MRS.py:208-213
        self.env.set_data(name, val)

    def get_data(self, name):
        return self.env.get_data(name)

    def __del__(self):   # no device work in a finaliser (check_errors() synchronises); close() is the explicit form
        pass

    def close(self):  # MRS.py:220-224: never raises
        try:
            self.check_errors()
        except Exception as exc:   # an unreported NaN action (lazy CHECK_NAN): the reference would have raised in step()
            warnings.warn("MRS.close(): %s" % (exc,), RuntimeWarning)

    def render(self, mode='bullet', close=False):  # MRS.py:227-229
        if close:
            self.close()

    def wait(self, dt=None):  # MRS.py:232-237
        if dt is None:
            dt = self.sim.DT
        diff = time.monotonic() - self.last_loop_time
        time.sleep(max(dt - diff, 0))

    def _poll_errors(self):
        """The in-loop form of check_errors(): never waits for the GPU.  Looks at the flag word the PREVIOUS poll copied to
        pinned host memory (its copy finished hundreds of steps ago), raises through check_errors() if it was set, and
        queues the next copy.  A NaN action therefore surfaces at most two polls (512 steps) after it was given -- or at
        the next reset()/reset_envs()/close(), which call check_errors() itself -- instead of stalling the stream for a
        synchronous read every 256 steps (~50 us of idle GPU per poll in the kernel trace)."""
        if getattr(self, "_err_host", None) is None:
            self._err_host = torch.zeros(1, dtype=torch.int32).pin_memory()
            self._err_event = None
        if self._err_event is not None and self._err_event.query():
            self._err_event = None
            if int(self._err_host[0]) & native.STATUS_NAN_ACTION:
                self.check_errors()
        if self._err_event is None:
            with torch.cuda.device(self.device):
                flag = (self.shard.status & native.STATUS_NAN_ACTION).max().reshape(1)
                self._err_host.copy_(flag, non_blocking=True)
                self._err_event = torch.cuda.Event()
                self._err_event.record()

    def check_errors(self):
        """Raise what the reference raises synchronously inside step(): NaN actions (MRS.py:247-248)."""
        st = self.shard.status
        if int((st & native.STATUS_NAN_ACTION).any()):
            bad = torch.nonzero(st & native.STATUS_NAN_ACTION).flatten().tolist()
            st.bitwise_and_(~native.STATUS_NAN_ACTION)
            raise Exception('The given action contains NaN (envs %s); those envs did not step' % bad)

    def step(self, actions, ACTION_TYPE=None):  # MRS.py:240-277
        E, N = self.N_ENVS, self.N_AGENTS
        atype = None
        if actions is not None:
            if not isinstance(actions, torch.Tensor):
                actions = torch.tensor(np.asarray(actions))
            if actions.requires_grad:
                actions = actions.detach()
            if actions.dim() == 1:
                actions = actions.reshape(self.N_AGENTS, self.ACTION_DIM)
            if (self.CHECK_NAN or ("sync" if E == 1 else "lazy")) == "sync" and bool(torch.isnan(actions).any()):
                raise Exception('The given action contains NaN:\n %s' % str(actions))
            self.last_action = actions
            atype = ACTION_TYPE if ACTION_TYPE is not None else self.ACTION_TYPE
            at = native.ACT.get(atype, -1) if atype is not None else -1
            if at <= 0:
                raise AttributeError("'Quadcopter' object has no attribute %r" % (atype,))  # Environment.py:92
            if actions.dtype != torch.float32 or actions.device != self.device or not actions.is_contiguous():
                actions = actions.to(device=self.device, dtype=torch.float32).contiguous()
            if actions.numel() != E * N * native.ACT_DIM[at]:
                raise ValueError("actions has %d elements, expected (%sN=%d, %d)" %
                                 (actions.numel(), "" if E == 1 else "E=%d, " % E, N, native.ACT_DIM[at]))
        else:
            at = 0
        # env.set_actions + sim.step_sim + newest X slice + newest A rows: one C-ABI call, raw slot pointers
        want_A = self.RETURN_A is None or bool(self.RETURN_A)
        fused = self._obs.fused
        xr, ar = self._Xring, self._Apacked
        xslot = xr.next_slot()
        aslot = ar.next_slot() if want_A else 0
        if self._dense_live and self._global_step - self._dense_read_step > self.DENSE_IDLE_STEPS:
            self._dense_live = False                # nobody has looked at info["A"] for a while: packed rows only again
        dr = self._Adense if (want_A and self._dense_live) else None   # A_FORMAT = "dense", being read: the float32 matrices come out of the same launch
        dslot = dr.next_slot() if dr is not None else 0
        self.shard.step_ptr(actions, at, xr.ptr(xslot) if fused else 0, ar.ptr(aslot) if want_A else 0,
                            float(self.COMM_RANGE), dr.ptr(dslot) if dr is not None else 0)
        if not fused:
            self._obs.write_into(xr.buf[xslot])
        xr.committed()
        Xk = self.get_Xk()
        Ak = None
        if want_A:
            ar.committed()
            if dr is not None:
                dr.committed()
            self._a_stamp += 1
            Ak = self.get_Ak()
        if (self._global_step & 255) == 255 and (self.CHECK_NAN or ("sync" if E == 1 else "lazy")) == "lazy":   # _global_step: never zeroed by reset()
            self._poll_errors()
        dc = self._default_cbs
        if self.update_fn is None and self.reward_fn is dc[0] and self.done_fn is dc[1] and self.info_fn is dc[2]:
            # every callback is the reference's default (MRS.py:38-40): reward 0.0, info {}, done = steps >= MAX_TIMESTEPS
            reward = 0.0
            self.last_obs = Xk
            info = {"A": Ak} if want_A else {}
            if self.RETURN_EVENTS:
                info["keyboard_events"] = self.env.get_keyboard_events()
                info["mouse_events"] = self.env.get_mouse_events()
            done = self.steps_since_reset >= self.MAX_TIMESTEPS
        else:
            # update function; draw_links is a GUI-only no-op here (MRS.py:259)
            kw = dict(env=self.env, X=Xk, A=Ak, action=self.last_action, steps_since_reset=self.steps_since_reset)
            if self.update_fn is not None:
                self.update_fn(Xlast=self.last_obs, **kw)
            reward = self.reward_fn(Xlast=self.last_obs, **kw)
            self.last_obs = Xk
            info = self.info_fn(Xlast=self.last_obs, **kw)   # sees Xlast == X, as upstream (MRS.py:264-266)
            if want_A:
                info["A"] = Ak
            if self.RETURN_EVENTS:
                info["keyboard_events"] = self.env.get_keyboard_events()
                info["mouse_events"] = self.env.get_mouse_events()
            done = self.done_fn(Xlast=self.last_obs, **kw)
        self.last_loop_time = time.monotonic()
        self.steps_since_reset += 1
        self._global_step += 1
        if self.AUTO_RESET:
            # gym vector-env convention: the observation returned for a finished env is its first one after the
            # reset; the terminal stack is kept in info["terminal_X"].  A tensor-valued `done` costs one host sync.
            if isinstance(done, torch.Tensor) and done.numel() == E and E > 1:
                dmask = done.to(self.device).to(torch.bool).reshape(E)
                if bool(dmask.any()):
                    info["terminal_X"], info["reset_mask"] = (Xk if self._copies() else Xk.clone()), dmask
                    Xk = self.reset_envs(dmask)
            elif bool(done):
                info["terminal_X"] = Xk if self._copies() else Xk.clone()
                Xk = self.reset()
        return Xk, reward, done, info

    def step_n(self, actions, n_substeps, ACTION_TYPE=None):
        """Frame skip (extension; the `n_substeps` of SURVEY.md 8b): `n_substeps` consecutive steps -- `actions` either one
        batch held for all of them, (E,N,adim), or one batch per substep, (S,E,N,adim) -- with the K_HOPS histories
        advanced every substep exactly as `n_substeps` step() calls would (bit-identical X / A), but one C call for the
        first S-1 of them and the callbacks (update / reward / info / done) run once, on the last substep, whose step()
        return value is returned.  A NaN action skips that substep of that env (flagged like in step())."""
        S = int(n_substeps)
        if S < 1:
            raise ValueError("n_substeps must be >= 1")
        if actions is None:
            for _ in range(S - 1):
                self._advance(None, 0, 1, 0)
            return self.step(None, ACTION_TYPE)
        E, N = self.N_ENVS, self.N_AGENTS
        a = actions if isinstance(actions, torch.Tensor) else torch.tensor(np.asarray(actions))
        atype = ACTION_TYPE if ACTION_TYPE is not None else self.ACTION_TYPE
        at = native.ACT.get(atype, -1) if atype is not None else -1
        if at <= 0:
            raise AttributeError("'Quadcopter' object has no attribute %r" % (atype,))
        a = a.detach().to(device=self.device, dtype=torch.float32).contiguous()
        per = E * N * native.ACT_DIM[at]
        if a.numel() == per:
            seq, stride = None, 0
        elif a.numel() == S * per:
            seq, stride = a.reshape(S, -1), per
        else:
            raise ValueError("actions has %d elements, expected (%sN=%d, %d) or %d of them" % (a.numel(), "" if E == 1 else "E=%d, " % E, N, native.ACT_DIM[at], S))
        if S > 1:
            self._advance(a if seq is None else seq[:S - 1], at, S - 1, stride)
        last = a if seq is None else seq[S - 1]
        return self.step(last.reshape((E, N, -1) if E > 1 else (N, -1)), ACTION_TYPE)

    def _advance(self, actions, at, n, act_stride):
        """n steps without callbacks: history rings advanced slot by slot (they grow towards slot 0, hence the negative
        strides), as many substeps per mrs_step_n call as fit before a ring wraps."""
        want_A = self.RETURN_A is None or bool(self.RETURN_A)
        xr, ar = self._Xring, self._Apacked
        fused = self._obs.fused
        done = 0
        while done < n:
            xslot = xr.next_slot()
            aslot = ar.next_slot() if want_A else 0
            cnt = 1
            if fused and xr.count > 0 and (not want_A or (ar.count > 0 and ar.head == xr.head and ar.L == xr.L)):
                cnt += min(n - done - 1, xr.head)            # slots below the new head that can be written before the wrap
            a_ptr = None
            if actions is not None:
                a_ptr = actions if act_stride == 0 else actions[done:]
            self.shard.step_n_ptr(a_ptr, at, cnt, act_stride, xr.ptr(xslot) if fused else 0, -xr._stride // 4,
                                  ar.ptr(aslot) if want_A else 0, -ar._stride // 8, float(self.COMM_RANGE))
            if not fused:
                self._obs.write_into(xr.buf[xslot])
            xr.committed()
            if want_A:
                ar.committed()
                self._expand_newest_A()
            for _ in range(cnt - 1):                          # the extra substeps of this call: slide the windows down
                xr.head -= 1
                xr.count = min(xr.count + 1, xr.K + 1)
                if want_A:
                    ar.head -= 1
                    ar.count = min(ar.count + 1, ar.K + 1)
                    self._expand_newest_A()
            done += cnt
        self.steps_since_reset += n
        self._global_step += n
        self._a_stamp += 1

    def get_env(self):  # MRS.py:280-293
        return self.env

    def get_objects(self):
        return self.env.objects

    def get_agents(self):
        return self.env.agents

    def get_controlled(self):
        return self.env.controlled

    def get_object_dict(self):
        return self.env.object_dict

    # ------------------------------------------------------------------ checkpoint
    def state_dict(self):
        return dict(shard=self.shard.state_dict(), steps_since_reset=self.steps_since_reset)

    def load_state_dict(self, sd):
        self.shard.load_state_dict(sd["shard"])
        self.steps_since_reset = sd["steps_since_reset"]
        self._clear_history()
        self.last_obs = self.calc_Xk()


_REGISTRY = {"mrs-v0": MRS}


def make(env_id='mrs-v0', **kwargs):
    """gym.make('mrs-v0', **kwargs) for installations without gym (mrsgym/__init__.py:13-16)."""
    if env_id not in _REGISTRY:
        raise KeyError("unknown environment id %r (registered: %s)" % (env_id, sorted(_REGISTRY)))
    return _REGISTRY[env_id](**kwargs)


def _register_with_gym():
    if _gym is None:
        return
    try:
        from gym.envs.registration import register  # type: ignore
    except Exception:
        try:
            from gymnasium.envs.registration import register  # type: ignore
        except Exception:
            return
    try:
        register(id='mrs-v0', entry_point='mrsgym_amd:MRS')
    except Exception:
        pass


_register_with_gym()
