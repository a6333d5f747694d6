"""Host-side logic of the product package on CPU tensors (no GPU, no compute calls)."""
import os

import numpy as np
import pytest
import torch

from mrsgym_amd.facade import StateFnCompiler
from mrsgym_amd.history import HistoryRing
from mrsgym_amd.util import CombinedDistribution, SphereTransform, randrange, totensor

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("K", [0, 1, 3])
@pytest.mark.parametrize("slots", [0, 2, 5])
def test_history_ring_reproduces_reference_deques(K, slots):
    """tests/golden/F4: calc_Xk pads with copies of X, calc_Ak with zeros, both newest-first (MRS.py:87-114)."""
    d = np.load(os.path.join(G, "F4_history.npz"))
    seqX, Xk, Ak = d["K%d_seqX" % K], d["K%d_Xk" % K], d["K%d_Ak" % K]
    T, N, D = seqX.shape
    xr = HistoryRing(K, (1, N, D), torch.float32, "cpu", slots=slots, pad="copy")
    ar = HistoryRing(K, (1, N, N), torch.float32, "cpu", slots=slots, pad="zero")
    seqP = d["K%d_seqP" % K]
    for rep in range(2):                      # second pass = after reset(): deques cleared
        xr.clear(); ar.clear()
        for t in range(T):
            xr.buf[xr.next_slot()][0] = torch.from_numpy(seqX[t])
            xr.committed()
            np.testing.assert_array_equal(xr.window()[:, 0].numpy(), Xk[t])
            if t >= 1:
                p = torch.from_numpy(seqP[t])
                A = ((p[:, None] - p[None]).norm(dim=2) <= 2.0).float()
                A.fill_diagonal_(0)
                ar.buf[ar.next_slot()][0] = A
                ar.committed()
                np.testing.assert_array_equal(ar.window()[:, 0].numpy(), Ak[t - 1])


def test_history_ring_long_run_wraps():
    K, L = 3, 9
    r = HistoryRing(K, (2,), torch.float32, "cpu", slots=L, pad="copy")
    for t in range(100):
        r.buf[r.next_slot()][:] = t
        r.committed()
        want = [max(t - k, 0) for k in range(K + 1)]
        assert r.window()[:, 0].tolist() == want
        assert r.window().data_ptr() == r.buf[r.head].data_ptr()       # a view, never a copy


class _FakeShard:
    E, N = 1, 3


def _recognise(fn):
    c = StateFnCompiler.__new__(StateFnCompiler)
    return c._recognise(fn)


def test_state_fn_recogniser():
    assert _recognise(lambda q: torch.cat([q.get_pos(), q.get_vel()])) == ("pos", "vel")           # README.md:28-29
    assert _recognise(lambda q: torch.cat([q.get_pos(), q.get_ori(), q.get_vel(), q.get_angvel()])) == \
        ("pos", "ori", "vel", "angvel")
    assert _recognise(lambda q: q.get_vel()) == ("vel",)
    assert _recognise(lambda q: torch.cat([q.get_vel(), q.get_pos()])) == ("vel", "pos")
    # not plain concatenations -> generic (vmap) path
    assert _recognise(lambda q: torch.cat([q.get_pos() * 2, q.get_vel()])) is None
    assert _recognise(lambda q: torch.cat([q.get_data("target")[q.get_idx()] - q.get_vel()])) is None  # magent.py:35-37
    assert _recognise(lambda q: torch.zeros(3)) is None
    assert _recognise(lambda q: torch.cat([q.get_pos(), q.get_ori(mat=True).flatten()])) is None


def test_default_spawn_distribution_properties():
    """Properties of MRS.default_spawn_dist (MRS.py:69-78) as captured in tests/golden/F5."""
    from torch.distributions import Normal, TransformedDistribution, Uniform
    d = np.load(os.path.join(G, "F5_spawn.npz"))
    for N in (3, 12, 32):
        ref = d["N%d" % N]
        assert (np.hypot(ref[..., 0], ref[..., 1]) <= 1 + 1e-6).all() and (ref[..., 2] >= 1).all() and (ref[..., 2] <= 3).all()
        dist = np.linalg.norm(ref[:, :, None] - ref[:, None], axis=-1) + np.eye(N) * 1e9
        assert dist.min() >= 0.6 - 1e-6
        z = Uniform(low=torch.ones(N, 1), high=3 * torch.ones(N, 1))
        xy = TransformedDistribution(Normal(torch.zeros(N, 2), 1.0), [SphereTransform(radius=1.0, within=True)])
        ours = CombinedDistribution([xy, z], mixer='cat', dim=1)
        s = torch.stack([ours.sample() for _ in range(200)])
        assert s.shape == (200, N, 3)
        r = s[..., :2].norm(dim=-1)
        assert float(r.max()) <= 1 + 1e-6 and 0.5 < float((r > 1 - 1e-6).float().mean()) < 0.72   # P(|N(0,I)|>1) = e^-0.5 = 0.607
        assert float(s[..., 2].min()) >= 1 and float(s[..., 2].max()) <= 3
    ori = d["ori_N16"]
    assert np.all(ori[..., :2] == 0) and np.abs(ori[..., 2]).max() <= np.pi / 2
    x = randrange(torch.tensor([0., 0., -1.]), torch.tensor([0., 0., 1.]))
    assert x.shape == (3,) and x[0] == 0 and abs(float(x[2])) <= 1
    assert totensor([1, 2]).tolist() == [1, 2]


def test_rollout_log_writes_the_reference_trainer_dict(tmp_path):
    """RolloutLog.save_trainer: the dict of deques Trainer.save_trainer writes (examples/simulating_data/helper/
    Trainer.py:43-61, :89-108) -- one (N,N) float32 A, (N,D) X, bool done, (N,3) expert and a context dict per
    sample, env after env in time order, the last sample of every env closing its episode.  (Checked once in the
    build container against the reference's own Trainer.load_trainer_dict / get_episodes / get_batch: 4 episodes,
    X (4,6,5,6) for this very input.)  Pure torch: runs on the CPU here, on device buffers in use."""
    import torch
    from collections import deque
    from mrsgym_amd.rollout import RolloutLog
    E, N, D, T = 3, 5, 6, 7
    log = RolloutLog(E, N, D, capacity=16, device="cpu")
    torch.manual_seed(0)
    Xs, As, Ex = [], [], []
    for t in range(T):
        X = torch.randn(E, N, D)
        A = ((torch.rand(E, N, N) > 0.5).float()) * (1 - torch.eye(N))
        ex = torch.randn(E, N, 3)
        if t % 2:       # packed rows in, as MRS(A_FORMAT="packed") hands them over
            packed = (A.to(torch.int64) << torch.arange(N)).sum(-1, keepdim=True)
            log.set_state(packed, X, done=torch.tensor([t == 3, False, False]), expert=ex)
        else:
            log.set_state(A, X, done=torch.tensor([t == 3, False, False]), expert=ex)
        Xs.append(X); As.append(A); Ex.append(ex)
    assert len(log) == T and torch.equal(log.dense_A(), torch.stack(As))
    path = str(tmp_path / "flocking.pt")
    log.save_trainer(path)
    data = torch.load(path, weights_only=False)        # a file this test wrote itself
    assert sorted(data) == ["history", "iter", "sample_idxs", "sample_weights"]
    h = data["history"]
    assert sorted(h) == ["A", "X", "context", "done", "expert"] and all(isinstance(v, deque) and len(v) == E * T for v in h.values())
    for e in range(E):
        for t in range(T):
            k = e * T + t
            assert torch.equal(h["X"][k], Xs[t][e]) and torch.equal(h["A"][k], As[t][e]) and torch.equal(h["expert"][k], Ex[t][e])
            assert h["A"][k].dtype == torch.float32 and h["context"][k] == {}
            assert h["done"][k] == (t == T - 1 or (e == 0 and t == 3))
    with pytest.raises(IndexError):
        for _ in range(20):
            log.set_state(As[0], Xs[0])


def test_combined_distribution_batched_sampling():
    """util.CombinedDistribution.sample(sample_shape): the mixing axis counts the axes of ONE sample (Util.py:86-99 has no
    sample_shape); MRS._draw batches a distribution that supports it and falls back to k calls for one that does not."""
    import torch
    from torch.distributions import Normal, Uniform
    from mrsgym_amd.util import CombinedDistribution
    from mrsgym_amd.mrs import MRS
    N = 5
    joint = CombinedDistribution([Normal(torch.zeros(N, 2), 1.25), Uniform(2.0 * torch.ones(N, 1), 5.0 * torch.ones(N, 1))], mixer='cat', dim=1)
    single = CombinedDistribution([Normal(torch.zeros(2), 1.25), Uniform(2.0 * torch.ones(1), 5.0 * torch.ones(1))], mixer='cat', dim=0)
    assert joint.sample().shape == (N, 3) and joint.sample((7,)).shape == (7, N, 3) and single.sample((4, 2)).shape == (4, 2, 3)
    x = MRS._draw(joint, 11)
    assert x.shape == (11, N, 3) and float(x[..., 2].min()) >= 2 and float(x[..., 2].max()) <= 5

    class OneAtATime:                       # e.g. the reference's own CombinedDistribution: sample() only
        def sample(self, *a):
            assert not a or a[0] == ()
            return torch.randn(3)
    assert MRS._draw(OneAtATime(), 6).shape == (6, 3)


def test_distribution_rebuilt_on_another_device():
    """MRS._dist_to: the START_POS distribution with its parameters on the device the swarm lives on (here 'cpu' stands in
    for it) -- torch's own distributions, TransformedDistribution, util.CombinedDistribution -- samples with the same
    shapes and the same support; what it cannot rebuild raises, and _spawn_user then samples the caller's object."""
    import pytest, torch
    from torch.distributions import Normal, Uniform, TransformedDistribution
    from mrsgym_amd.util import CombinedDistribution, SphereTransform
    from mrsgym_amd.mrs import MRS
    N = 6
    xy = TransformedDistribution(Normal(torch.zeros(N, 2), 1.0), [SphereTransform(radius=1.0, within=True)])
    d = CombinedDistribution([xy, Uniform(1.0 * torch.ones(N, 1), 3.0 * torch.ones(N, 1))], mixer='cat', dim=1)
    e = MRS._dist_to(d, torch.device("cpu"))
    assert e is not d and e.dist[0].base_dist is not d.dist[0].base_dist
    x = e.sample((50,))
    assert x.shape == (50, N, 3) and float(x[..., 2].min()) >= 1 and float(x[..., 2].max()) <= 3
    assert float(x[..., :2].norm(dim=-1).max()) <= 1 + 1e-6           # the transform travelled with it

    class Opaque:
        def sample(self, *a):
            return torch.randn(3)
    with pytest.raises(TypeError):
        MRS._dist_to(Opaque(), torch.device("cpu"))
