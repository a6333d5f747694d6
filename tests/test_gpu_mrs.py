"""The drop-in Gym surface (mrsgym_amd.MRS / make('mrs-v0')) on the GPU, read like the reference's
README / examples would exercise it, and checked against the reference's own MRS.step() outputs
(tests/golden/F6, fake-bullet harness)."""
import glob
import os

import numpy as np
import pytest

import oracle
from util_scenarios import ActionStream, grid_spawn

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def state_fn(quad):
    return torch.cat([quad.get_pos(), quad.get_vel()])


def state_fn_full(quad):
    return torch.cat([quad.get_pos(), quad.get_ori(), quad.get_vel(), quad.get_angvel()])


def test_readme_example_plumbing():
    """README.md:14-33: N=3, set_target_vel, state_fn = cat(pos, vel) (BASELINE config 1)."""
    import mrsgym_amd
    N = 3
    env = mrsgym_amd.make('mrs-v0', N_AGENTS=N, state_fn=state_fn, ACTION_TYPE='set_target_vel')
    assert env._obs.fused and env._obs.fields == ("pos", "vel")
    for _ in range(20):
        actions = torch.tensor([0.5, 0.0, 0.0]).expand(N, -1)
        X, reward, done, info = env.step(actions)
    assert X.shape == (1, N, 6) and X.dtype == torch.float32           # (K+1, N, D), MRS.py:99
    assert info["A"].shape == (1, N, N) and reward == 0.0 and done is False
    assert torch.equal(info["A"][0].cpu(), torch.ones(N, N) - torch.eye(N))   # COMM_RANGE = inf, MRS.py:118-119
    assert float(X[0, :, 3].mean()) > 0.0                               # moving along +x
    env.wait(0.0)
    env.close()


def test_default_spawn_and_reset_semantics():
    import mrsgym_amd
    N = 12
    env = mrsgym_amd.make('mrs-v0', N_AGENTS=N, state_fn=state_fn, K_HOPS=2, COMM_RANGE=2.5, SEED=3)
    X = env.reset()
    assert X.shape == (3, N, 6)
    assert torch.equal(X[0], X[1]) and torch.equal(X[0], X[2])         # padded with copies (MRS.py:92-93)
    p = X[0, :, :3].cpu().numpy()
    assert (np.hypot(p[:, 0], p[:, 1]) <= 1 + 1e-6).all() and (p[:, 2] >= 1).all() and (p[:, 2] <= 3).all()
    d = np.linalg.norm(p[:, None] - p[None], axis=-1) + np.eye(N) * 1e9
    assert d.min() >= 0.6 - 1e-6
    assert torch.all(X[0, :, 3:] == 0)
    X2 = env.reset()
    assert not torch.equal(X, X2)                                       # re-sampled
    Xs, r, dn, info = env.step(torch.zeros(N, 3))
    A = info["A"]
    assert A.shape == (3, N, N) and float(A[1:].abs().sum()) == 0      # A history padded with zeros (MRS.py:107-108)
    Xs2, _, _, info2 = env.step(torch.zeros(N, 3))
    assert torch.equal(info2["A"][1], A[0]) and torch.equal(Xs2[1], Xs[0])
    # explicit pos / vel override (README.md:82-95); set() keeps what is not given (MRS.py:196-205)
    pos = torch.tensor(grid_spawn(1, N)[0][0])
    X3 = env.reset(pos=pos, vel=torch.ones(N, 3))
    np.testing.assert_array_equal(X3[0, :, :3].cpu().numpy(), pos.numpy())
    assert torch.all(X3[0, :, 3:] == 1)
    X4 = env.set(vel=torch.zeros(N, 3))
    np.testing.assert_array_equal(X4[0, :, :3].cpu().numpy(), pos.numpy())
    assert torch.all(X4[0, :, 3:] == 0) and env.steps_since_reset == 0
    # N = 64 does not fit the default volume: the reference spins forever, we raise
    with pytest.raises(RuntimeError):
        mrsgym_amd.make('mrs-v0', N_AGENTS=64, state_fn=state_fn)


def test_user_spawn_distribution_and_start_ori():
    """START_POS as a torch distribution (gen_data.py:30-36 style) and per-agent START_ORI ranges."""
    import mrsgym_amd
    from torch.distributions import Normal, Uniform
    from mrsgym_amd.util import CombinedDistribution
    N = 12
    z = Uniform(low=2.0 * torch.ones(N, 1), high=5.0 * torch.ones(N, 1))
    xy = Normal(torch.zeros(N, 2), 1.25)
    joint = CombinedDistribution([xy, z], mixer='cat', dim=1)                   # (N,3) joint samples
    single = CombinedDistribution([Normal(torch.zeros(2), 1.25), Uniform(2.0 * torch.ones(1), 5.0 * torch.ones(1))],
                                  mixer='cat', dim=0)                            # (3,) per-agent samples
    for dist_, E in ((joint, 1), (single, 1), (joint, 3)):
        env = mrsgym_amd.make('mrs-v0', N_ENVS=E, N_AGENTS=N, state_fn=state_fn, START_POS=dist_,
                              START_ORI=torch.tensor([0., 0., -1., 0., 0., 1.]))
        X = env.reset()
        p = (X if E > 1 else X.unsqueeze(0))[:, 0, :, :3].cpu().numpy()
        d = np.linalg.norm(p[:, :, None] - p[:, None], axis=-1) + np.eye(N) * 1e9
        assert d.min() >= 0.6 - 1e-6 and (p[..., 2] >= 2).all() and (p[..., 2] <= 5).all()
        yaw = env.get_env().get_ori()[..., 2].abs().max()
        assert float(yaw) <= 1.0 + 1e-6


def test_user_spawn_is_batched_on_the_device():
    """START_POS as a distribution at E = 4096, N = 12: one bulk draw + mrs_spawn_from, no per-env host
    loop -- under 50 ms per reset() after the first; reset_envs() re-draws the selected envs only and moves only those."""
    import time
    import mrsgym_amd
    from torch.distributions import Normal, Uniform
    from mrsgym_amd.util import CombinedDistribution
    E, N = 4096, 12
    dist_ = CombinedDistribution([Normal(torch.zeros(N, 2), 1.25), Uniform(2.0 * torch.ones(N, 1), 5.0 * torch.ones(N, 1))], mixer='cat', dim=1)
    env = mrsgym_amd.make('mrs-v0', N_ENVS=E, N_AGENTS=N, state_fn=state_fn, START_POS=dist_)
    dt = 1e9
    for _ in range(5):                     # best of five: the box's host share is 16 busy cores, one slow call happens
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        X = env.reset()
        torch.cuda.synchronize()
        dt = min(dt, time.perf_counter() - t0)
    p = X[:, 0, :, :3]
    d = torch.cdist(p, p) + 10 * torch.eye(N, device=p.device)
    assert float(d.min()) >= 0.6 - 1e-6 and float(p[..., 2].min()) >= 2 and float(p[..., 2].max()) <= 5
    assert float(p[..., :2].std()) > 1.0                                   # the Normal(0, 1.25) cloud, not a degenerate layout
    assert dt < 0.05, "reset() took %.1f ms" % (dt * 1e3)
    assert env._sp_cache[1] is not None and env._sp_cache[1].dist[0].loc.is_cuda    # sampled on the device (MRS._dist_to)
    before = env.shard.view(env.shard.pos).clone()
    mask = torch.zeros(E, dtype=torch.bool); mask[::7] = True
    env.reset_envs(mask)
    after = env.shard.view(env.shard.pos)
    assert torch.equal(after[~mask.cuda()], before[~mask.cuda()]) and not torch.equal(after[mask.cuda()], before[mask.cuda()])
    d = torch.cdist(after.float(), after.float()) + 10 * torch.eye(N, device=p.device)
    assert float(d.min()) >= 0.6 - 1e-6
    # per-agent (3,) samples, and a layout that needs many rounds (12 agents of radius 0.3 in a tight cloud)
    single = CombinedDistribution([Normal(torch.zeros(2), 0.8), Uniform(2.0 * torch.ones(1), 3.0 * torch.ones(1))], mixer='cat', dim=0)
    env2 = mrsgym_amd.make('mrs-v0', N_ENVS=64, N_AGENTS=N, state_fn=state_fn, START_POS=single)
    p2 = env2.get_Xk()[:, 0, :, :3]
    assert float((torch.cdist(p2, p2) + 10 * torch.eye(N, device=p2.device)).min()) >= 0.6 - 1e-6

    class Opaque:                          # a caller's own sampler: nothing to rebuild on the device -> drawn on the host
        def sample(self, shape=()):
            return single.sample(shape)
    env3 = mrsgym_amd.make('mrs-v0', N_ENVS=64, N_AGENTS=N, state_fn=state_fn, START_POS=Opaque())
    assert env3._sp_cache[1] is None
    m3 = torch.zeros(64, dtype=torch.bool); m3[5] = True
    b3 = env3.shard.view(env3.shard.pos).clone()
    env3.reset_envs(m3)
    p3 = env3.shard.view(env3.shard.pos)
    assert torch.equal(p3[~m3.cuda()], b3[~m3.cuda()]) and not torch.equal(p3[5], b3[5])
    assert float((torch.cdist(p3.float(), p3.float()) + 10 * torch.eye(N, device=p3.device)).min()) >= 0.6 - 1e-6


def test_callbacks_and_quirks():
    import mrsgym_amd
    N = 5
    seen = {}

    def reward_fn(env, X, A, Xlast, action, steps_since_reset):
        seen["reward"] = (X.shape, A.shape, Xlast is not X, steps_since_reset, env.get_pos().shape)
        return -float(X[0, :, 2].mean())

    def info_fn(**kw):
        seen["info_xlast_is_x"] = kw["Xlast"] is kw["X"]               # MRS.py:264-266 quirk
        return {"t": kw["steps_since_reset"]}

    def update_fn(**kw):
        seen["update"] = True

    def start_fn(gymenv):
        gymenv.set_data("target_vel", torch.randn(gymenv.N_AGENTS, 3, device=gymenv.device))

    def generic_state_fn(quad):                                          # magent.py:35-37: not fusable
        tv = quad.get_data("target_vel")[quad.get_idx(), :]
        return torch.cat([tv - quad.get_vel()])

    pos = torch.tensor(grid_spawn(1, N)[0][0])
    env = mrsgym_amd.make('mrs-v0', N_AGENTS=N, state_fn=generic_state_fn, reward_fn=reward_fn, info_fn=info_fn,
                          update_fn=update_fn, start_fn=start_fn, MAX_TIMESTEPS=3, START_POS=pos, COMM_RANGE=1.5)
    assert not env._obs.fused
    X = env.reset()
    assert X.shape == (1, N, 3)
    np.testing.assert_allclose(X[0].cpu().numpy(), env.get_data("target_vel").cpu().numpy(), atol=1e-7)
    dones = []
    for t in range(4):
        X, r, d, info = env.step(torch.zeros(N, 3))
        dones.append(d)
        assert info["t"] == t and isinstance(r, float)
    assert dones == [False, False, False, True]                         # steps_since_reset >= MAX_TIMESTEPS before the increment
    assert seen["reward"][0] == (1, N, 3) and seen["reward"][1] == (1, N, N) and seen["reward"][4] == (N, 3)
    assert seen["info_xlast_is_x"] and seen["update"]
    # step(None) skips the actions entirely (MRS.py:243-253); unknown ACTION_TYPE -> AttributeError
    env.step(None)
    with pytest.raises(AttributeError):
        env.step(torch.zeros(N, 3), ACTION_TYPE="set_force")            # README.md:68 has no implementation
    with pytest.raises(Exception, match="NaN"):
        env.step(torch.full((N, 3), float("nan")))                      # MRS.py:247-248
    # per-agent views
    q = env.get_agents()[2]
    assert q.get_idx() == 2 and q.get_pos().shape == (3,) and q.get_ori().shape == (3,) and q.get_ori(mat=True).shape == (3, 3)
    np.testing.assert_allclose(q.get_pos().cpu().numpy(), env.get_env().get_pos()[2].cpu().numpy())
    assert env.get_env().draw_links(None) is None and env.get_env().set_colour(q, [1, 0, 0]) is None


def test_lazy_nan_action_is_raised_even_with_short_episodes():
    """CHECK_NAN="lazy" (the default for N_ENVS > 1): a NaN action is flagged on the device, the env does not step
    (MRS.py:247-248) and the exception surfaces at the next poll -- which must not depend on episodes being
    longer than the poll period, nor be wiped by reset()'s spawn (episodes of 50 steps here)."""
    import mrsgym_amd
    E, N = 4, 6
    env = mrsgym_amd.make('mrs-v0', N_ENVS=E, N_AGENTS=N, state_fn=state_fn, SEED=5)
    raised_at = None
    a = torch.zeros(E, N, 3, device="cuda")
    try:
        for t in range(600):
            if t % 50 == 0 and t > 0:
                env.reset()
            act = a.clone()
            if t == 3:
                act[2, 1, 0] = float("nan")
            env.step(act)
    except Exception as exc:
        raised_at = t
        assert "NaN" in str(exc) and "[2]" in str(exc)
    assert raised_at is not None and raised_at <= 50      # the reset() after the first episode at the latest
    # ... and the flag is consumed: the loop can go on
    env.reset()
    for t in range(5):
        env.step(a)
    env.check_errors()


def test_lazy_nan_poll_never_blocks_and_still_raises():
    """No reset in the loop: the in-loop poll (every 256 steps) copies the flag word to pinned memory without waiting and
    looks at it one poll later, so the exception surfaces within three poll periods."""
    import mrsgym_amd
    E, N = 4, 6
    env = mrsgym_amd.make('mrs-v0', N_ENVS=E, N_AGENTS=N, state_fn=state_fn, SEED=5, MAX_TIMESTEPS=10**9)
    a = torch.zeros(E, N, 3, device="cuda")
    raised_at = None
    try:
        for t in range(800):
            act = a.clone()
            if t == 3:
                act[1, 0, 2] = float("nan")
            env.step(act)
    except Exception as exc:
        raised_at = t
        assert "NaN" in str(exc) and "[1]" in str(exc)
    assert raised_at is not None and raised_at <= 767


def test_generic_state_fn_is_vmapped_across_envs():
    """A state_fn the recogniser cannot fuse (arithmetic on the getters, per-agent data), N_ENVS > 1: evaluated for all
    E*N quadcopters at once with torch.func.vmap -- same numbers as the per-agent formula on the batched getters."""
    import mrsgym_amd
    E, N = 5, 7
    pos, eul = grid_spawn(E, N, seed=6)

    def fn(quad):
        gain = quad.get_data("gain")[quad.get_idx()]
        return torch.cat([quad.get_pos() * 2.0 - quad.get_vel(), quad.get_ori()[2:3] * gain, quad.get_angvel().norm().reshape(1)])
    env = mrsgym_amd.make('mrs-v0', N_ENVS=E, N_AGENTS=N, state_fn=fn, K_HOPS=1, COMM_RANGE=2.0,
                          START_POS=torch.from_numpy(pos), start_fn=lambda m: m.set_data("gain", torch.arange(1., N + 1, device=m.device)))
    assert not env._obs.fused
    X = env.reset(ori=torch.from_numpy(eul))
    assert X.shape == (E, 2, N, 5)
    acts = ActionStream("set_target_vel", E, N, pos, seed=2)
    for t in range(12):
        X, r, d, info = env.step(torch.from_numpy(acts(t)))
    w = env.get_env()
    gain = torch.arange(1., N + 1, device="cuda")
    want = torch.cat([w.get_pos() * 2.0 - w.get_vel(), w.get_ori()[..., 2:3] * gain[None, :, None],
                      w.get_angvel().norm(dim=-1, keepdim=True)], -1)
    assert torch.allclose(X[:, 0], want, rtol=0, atol=1e-6)
    assert float(X[:, 0, :, 4].abs().max()) > 0


F6 = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "F6_step_N*.npz")))


@pytest.mark.parametrize("path", F6, ids=[os.path.basename(p)[8:-4] for p in F6])
def test_reference_mrs_step_outputs(path):
    """Drive our MRS with the reference run's inputs; X / A / reward / done must match the reference's
    own step() outputs while the trajectory is still in its pre-chaotic window (first 40 steps)."""
    import mrsgym_amd
    d = np.load(path)
    name = os.path.basename(path)[8:-4]
    N, atype = int(name.split("_")[0][1:]), name.split("_", 1)[1]
    D, K, cr = int(d["D"]), int(d["K_HOPS"]), float(d["COMM_RANGE"])
    env = mrsgym_amd.make('mrs-v0', state_fn=state_fn_full if D == 12 else state_fn, N_AGENTS=N, K_HOPS=K, COMM_RANGE=cr,
                          ACTION_TYPE=atype, HEADLESS=True, START_POS=torch.tensor(d["start"]))
    X0 = env.reset(ori=torch.tensor(d["ori0"]))
    np.testing.assert_allclose(X0.cpu().numpy(), d["X0"], rtol=0, atol=1e-6)
    for t in range(40):
        X, r, dn, info = env.step(torch.tensor(d["actions"][t]))
        np.testing.assert_allclose(X.cpu().numpy(), d["X"][t], rtol=0, atol=2e-5, err_msg="%s t=%d" % (name, t))
        mism = (info["A"].cpu().numpy().astype(np.uint8) != d["A"][t]).sum()
        assert mism == 0, (name, t, mism)
        assert r == d["reward"][t] and dn == bool(d["done"][t])


def test_default_dense_A_is_lazy_and_identical():
    """the default construction (A_FORMAT = "dense") returns info["A"] as a torch.Tensor of the reference's shape and
    dtype that is materialised on first use (mrsgym_amd/lazy.py).  (a) read every step it equals the expansion of the packed rows of
    a twin environment bit for bit (K_HOPS history, zero padding after reset, a masked reset in between); (b) never read, no float32
    ring exists at all; (c) read once, the step kernel keeps the ring current until nobody has looked for DENSE_IDLE_STEPS steps; (d) a
    stack read late comes from its own rows: the clone it kept (copies, N_ENVS = 1) or its ring slots while they last (views; an error after)."""
    import mrsgym_amd
    from mrsgym_amd.lazy import LazyDenseA
    E, N, K = 3, 64, 2
    pos, _ = grid_spawn(E, N, seed=11)
    acts = ActionStream("set_target_vel", E, N, pos, seed=12)

    def mk(fmt, **kw):
        env = mrsgym_amd.make('mrs-v0', N_ENVS=E, N_AGENTS=N, state_fn=state_fn, K_HOPS=K, COMM_RANGE=2.0, ACTION_TYPE="set_target_vel",
                              START_POS=torch.from_numpy(pos), HEADLESS=True, A_FORMAT=fmt, **kw)
        env.reset(ori=torch.zeros(E, N, 3))      # (the default START_ORI draws a yaw per agent: the twins must start alike)
        return env

    def expand(env, packed):            # (E, K+1, N, W) int64 -> (E, K+1, N, N) float32
        p = packed.contiguous()
        out = torch.empty(p.shape[0] * p.shape[1], N, N, device=p.device)
        env.shard.adjacency_expand(p.view(-1, N, p.shape[-1]), out)
        return out.view(p.shape[0], p.shape[1], N, N)

    dense, packed, unread = mk("dense"), mk("packed"), mk("dense")
    for t in range(40):
        a = torch.from_numpy(acts(t)).cuda()
        if t == 17:
            m = torch.tensor([True, False, True], device="cuda")
            for env in (dense, packed, unread):
                env.reset_envs(m, ori=torch.zeros(E, N, 3))
        A = dense.step(a)[3]["A"]
        Ap = packed.step(a)[3]["A"]
        unread.step(a)
        assert isinstance(A, torch.Tensor) and isinstance(A, LazyDenseA) and A.shape == (E, K + 1, N, N) and A.dtype == torch.float32 and A.device.type == "cuda"
        assert torch.equal(A, expand(packed, Ap)), t                       # (a)
    assert unread._Adense is None and not unread._dense_live                # (b)
    assert dense._dense_live
    for t in range(40, 40 + dense.DENSE_IDLE_STEPS + 3):                    # (c)
        a = torch.from_numpy(acts(t)).cuda()
        last = dense.step(a)[3]["A"]
        packed.step(a)
    assert not dense._dense_live
    assert torch.equal(last.materialize(), expand(packed, packed.get_Ak())) and dense._dense_live   # ... and comes back on the next read
    a0 = torch.from_numpy(acts(0)).cuda()
    stale = dense.step(a0)[3]["A"]                                           # (d) views: a stack first read after the env has moved on ...
    stale_p = packed.step(a0)[3]["A"].clone()
    dense.step(a0); packed.step(a0)
    assert torch.equal(stale, expand(packed, stale_p))                        # ... still stands for its own step (the reference's data loops log the previous A)
    gone = dense.step(a0)[3]["A"]
    for _ in range(dense._Apacked.L):                                         # ... until its ring slots have been reused
        dense.step(a0)
    with pytest.raises(RuntimeError):
        gone.sum()
    one = mrsgym_amd.make('mrs-v0', N_AGENTS=N, state_fn=state_fn, K_HOPS=K, COMM_RANGE=2.0, START_POS=torch.from_numpy(pos[0]), HEADLESS=True)
    one.reset(ori=torch.zeros(N, 3))
    ref = mrsgym_amd.make('mrs-v0', N_AGENTS=N, state_fn=state_fn, K_HOPS=K, COMM_RANGE=2.0, START_POS=torch.from_numpy(pos[0]), HEADLESS=True, A_FORMAT="packed")
    ref.reset(ori=torch.zeros(N, 3))
    a1 = torch.from_numpy(acts(0)[0])
    old = one.step(a1)[3]["A"]
    oldp = ref.step(a1)[3]["A"]
    one.step(a1); ref.step(a1)
    assert old.shape == (K + 1, N, N) and torch.equal(old, expand(ref, oldp[None])[0])      # copies: its own rows, whatever happened since


def test_vectorised_envs_match_single_env_runs():
    """N_ENVS = E adds a leading axis and nothing else: env e of the batch == a single-env run of env e."""
    import mrsgym_amd
    E, N, K = 3, 12, 1
    pos, eul = grid_spawn(E, N, yaw_range=0.8)
    acts = ActionStream("set_target_vel", E, N, pos, seed=4, coherent=True)
    batch = mrsgym_amd.make('mrs-v0', N_ENVS=E, N_AGENTS=N, state_fn=state_fn, K_HOPS=K, COMM_RANGE=2.5,
                            START_POS=torch.tensor(pos))
    batch.reset(ori=torch.tensor(eul))
    singles = []
    for e in range(E):
        s = mrsgym_amd.make('mrs-v0', N_AGENTS=N, state_fn=state_fn, K_HOPS=K, COMM_RANGE=2.5, START_POS=torch.tensor(pos[e]))
        s.reset(ori=torch.tensor(eul[e]))
        singles.append(s)
    for t in range(30):
        a = torch.tensor(acts(t))
        Xb, rb, db, ib = batch.step(a)
        assert Xb.shape == (E, K + 1, N, 6) and ib["A"].shape == (E, K + 1, N, N)
        for e, s in enumerate(singles):
            Xs, _, _, is_ = s.step(a[e])
            assert torch.equal(Xs, Xb[e]) and torch.equal(is_["A"], ib["A"][e])
    # packed adjacency format carries the same bits
    pk = mrsgym_amd.make('mrs-v0', N_ENVS=E, N_AGENTS=N, state_fn=state_fn, K_HOPS=K, COMM_RANGE=2.5,
                         START_POS=torch.tensor(pos), A_FORMAT="packed", RETURN_A=True)
    pk.reset(ori=torch.tensor(eul))
    acts = ActionStream("set_target_vel", E, N, pos, seed=4, coherent=True)
    b2 = mrsgym_amd.make('mrs-v0', N_ENVS=E, N_AGENTS=N, state_fn=state_fn, K_HOPS=K, COMM_RANGE=2.5, START_POS=torch.tensor(pos))
    b2.reset(ori=torch.tensor(eul))
    for t in range(5):
        a = torch.tensor(acts(t))
        _, _, _, ip = pk.step(a)
        _, _, _, idn = b2.step(a)
    bits = ip["A"][..., 0]                                              # (E,K+1,N) int64 rows, N <= 64
    dense = ((bits.unsqueeze(-1) >> torch.arange(N, device=bits.device)) & 1).float()
    assert torch.equal(dense, idn["A"])
    # RETURN_A=False skips the adjacency (README.md:62 semantics; upstream ignores the flag)
    na = mrsgym_amd.make('mrs-v0', N_AGENTS=N, state_fn=state_fn, START_POS=torch.tensor(pos[0]), RETURN_A=False)
    _, _, _, info = na.step(torch.zeros(N, 3))
    assert "A" not in info


def test_step_n_is_n_steps_bit_for_bit():
    """MRS.step_n / mrs_step_n (the n_substeps of SURVEY.md 8b): S substeps from one call == S step() calls -- state, the
    K_HOPS observation and adjacency histories (incl. across a history-ring wrap), dense and packed A, held and per-substep
    actions; the callbacks run once, on the last substep."""
    import mrsgym_amd
    E, N, K = 3, 12, 2
    pos, eul = grid_spawn(E, N, yaw_range=0.8)
    calls = {"n": 0}

    def reward_fn(**kw):
        calls["n"] += 1
        return 0.0
    for fmt in ("dense", "packed"):
        kw = dict(N_ENVS=E, N_AGENTS=N, state_fn=state_fn, K_HOPS=K, COMM_RANGE=2.5, START_POS=torch.from_numpy(pos), A_FORMAT=fmt,
                  ACTION_TYPE="set_target_vel", HISTORY_SLOTS=7)
        a_env = mrsgym_amd.make('mrs-v0', reward_fn=reward_fn, **kw)
        b_env = mrsgym_amd.make('mrs-v0', **kw)
        for e_ in (a_env, b_env):
            e_.reset(ori=torch.from_numpy(eul))
        acts = ActionStream("set_target_vel", E, N, pos, seed=8, coherent=True)
        t = 0
        for S, per_substep in ((1, False), (4, False), (5, True), (9, True), (3, False)):       # 22 substeps over a 7-slot ring: several wraps
            seq = torch.from_numpy(np.stack([acts(t + s) for s in range(S)])).cuda()
            calls["n"] = 0
            Xa, ra, da, ia = a_env.step_n(seq if per_substep else seq[0], S)
            assert calls["n"] == 1
            for s in range(S):
                Xb, rb, db, ib = b_env.step(seq[s] if per_substep else seq[0])
            assert torch.equal(Xa, Xb) and torch.equal(ia["A"], ib["A"]), (fmt, S)
            for name in ("pos", "quat", "vel", "angvel", "pid"):
                assert torch.equal(torch.nan_to_num(getattr(a_env.shard, name)), torch.nan_to_num(getattr(b_env.shard, name))), name
            assert a_env.steps_since_reset == b_env.steps_since_reset
            t += S
    with pytest.raises(ValueError):
        a_env.step_n(torch.zeros(2, E, N, 3), 3)


def test_checkpoint_roundtrip():
    import mrsgym_amd
    N = 6
    pos = torch.tensor(grid_spawn(1, N)[0][0])
    env = mrsgym_amd.make('mrs-v0', N_AGENTS=N, state_fn=state_fn, START_POS=pos)
    for t in range(10):
        env.step(torch.full((N, 3), 0.2))
    sd = env.state_dict()
    a = [env.step(torch.full((N, 3), -0.1))[0].clone() for _ in range(5)]
    env.load_state_dict(sd)
    b = [env.step(torch.full((N, 3), -0.1))[0].clone() for _ in range(5)]
    assert all(torch.equal(x, y) for x, y in zip(a, b))


def test_reset_envs_touches_only_the_selected_envs():
    """Per-env reset (vectorised extension): the selected envs behave like a fresh MRS.reset (MRS.py:174-192: new
    state, zero velocities, X history padded with the new observation, A history empty), the others keep state,
    controller memory and history bit for bit."""
    import mrsgym_amd
    E, N, K = 4, 8, 2
    pos, eul = grid_spawn(E, N, seed=3)
    kw = dict(N_ENVS=E, N_AGENTS=N, state_fn=state_fn, K_HOPS=K, COMM_RANGE=2.0, START_POS=torch.from_numpy(pos),
              ACTION_TYPE="set_target_vel")
    env, ref = mrsgym_amd.make('mrs-v0', **kw), mrsgym_amd.make('mrs-v0', **kw)
    for e_ in (env, ref):
        e_.reset(ori=torch.from_numpy(eul))
    acts = ActionStream("set_target_vel", E, N, pos, seed=5, coherent=True)
    for t in range(15):
        a = torch.from_numpy(acts(t)).cuda()
        env.step(a); ref.step(a)
    mask = torch.tensor([False, True, False, True])
    newpos = torch.from_numpy(pos) + torch.tensor([0.1, -0.2, 0.5])
    Xk = env.reset_envs(mask, pos=newpos, ori=torch.from_numpy(eul))
    Xr = ref.get_Xk()
    assert Xk.shape == (E, K + 1, N, 6)
    keep = ~mask
    assert torch.equal(Xk[keep.cuda()], Xr[keep.cuda()])                          # untouched envs: history intact
    for k in range(K + 1):                                                         # reset envs: padded with the new X
        assert torch.allclose(Xk[mask.cuda()][:, k, :, :3].cpu(), newpos[mask].float(), atol=1e-6)
        assert float(Xk[mask.cuda()][:, k, :, 3:].abs().max()) == 0.0
    for name in ("pos", "quat", "vel", "angvel", "pid"):
        a, b = getattr(env.shard, name), getattr(ref.shard, name)
        va, vb = a.view(a.shape[0], E, N, -1), b.view(b.shape[0], E, N, -1)    # planes of scalars, or of 16-byte records (pid)
        assert torch.equal(va[:, keep.cuda()], vb[:, keep.cuda()]), name          # state + controller memory intact
    assert torch.equal(env.env_steps().cpu(), torch.tensor([15, 0, 15, 0]))
    # next step: A history of the reset envs is [A_new, 0, 0]; the others carry on exactly like the reference run
    a = torch.from_numpy(acts(15)).cuda()
    X1, _, _, info1 = env.step(a)
    X2, _, _, info2 = ref.step(a)
    assert torch.equal(X1[keep.cuda()], X2[keep.cuda()]) and torch.equal(info1["A"][keep.cuda()], info2["A"][keep.cuda()])
    A = info1["A"][mask.cuda()]
    assert float(A[:, 1:].abs().max()) == 0.0 and float(A[:, 0].sum()) > 0
    # default spawn through the env mask: only the selected envs move, min separation holds (MRS.py:137-153)
    env2 = mrsgym_amd.make('mrs-v0', N_ENVS=E, N_AGENTS=N, state_fn=state_fn, K_HOPS=1)
    before = env2.shard.view(env2.shard.pos).clone()
    env2.reset_envs(mask)
    after = env2.shard.view(env2.shard.pos)
    assert torch.equal(after[keep.cuda()], before[keep.cuda()]) and not torch.equal(after[mask.cuda()], before[mask.cuda()])
    d = torch.cdist(after[mask.cuda()].float(), after[mask.cuda()].float()) + 10 * torch.eye(N, device="cuda")
    assert float(d.min()) >= 0.6 - 1e-6


def test_auto_reset_on_done_mask():
    """AUTO_RESET: envs flagged by a tensor-valued done_fn restart inside step(); the returned stack is their first
    observation after the reset, the terminal one is in info (gym vector-env convention)."""
    import mrsgym_amd
    E, N = 3, 4
    pos, eul = grid_spawn(E, N, seed=1)

    def done_fn(X=None, **kw):   # an env is done once its first agent has climbed above z = 2.2
        return X[:, 0, 0, 2] > 2.2
    env = mrsgym_amd.make('mrs-v0', N_ENVS=E, N_AGENTS=N, state_fn=state_fn, K_HOPS=1, START_POS=torch.from_numpy(pos),
                          ACTION_TYPE="set_target_vel", done_fn=done_fn, AUTO_RESET=True)
    start_z = env.get_Xk()[:, 0, 0, 2].clone()
    up = torch.zeros(E, N, 3, device="cuda"); up[0, :, 2] = 2.0; up[2, :, 2] = 0.5     # env 0 climbs fast, env 1 hovers
    fired = None
    for t in range(400):
        X, r, done, info = env.step(up)
        if "reset_mask" in info:
            fired = (t, info["reset_mask"].cpu().tolist(), info["terminal_X"], X)
            break
    assert fired is not None and fired[1][0] and not fired[1][1]
    t, m, Xterm, X = fired
    assert float(Xterm[0, 0, 0, 2]) > 2.2 and abs(float(X[0, 0, 0, 2]) - float(start_z[0])) < 1e-6
    assert torch.equal(X[0, 0], X[0, 1])                       # history of the restarted env is padded with its new X
    assert env.env_steps().cpu().tolist()[0] == 0 and env.env_steps().cpu().tolist()[1] == t + 1


def test_reynolds_expert_kernel(golden_dir):
    """mrs_reynolds: bit-identical to the oracle restatement, within 2e-6 of the reference's own
    forward_batch outputs (tests/golden/F7), in the reference's input layout, this library's stack layout and
    straight from an env's history ring."""
    import mrsgym_amd
    d = np.load(os.path.join(golden_dir, "F7_reynolds.npz"))
    for k in d.files:
        if not k.endswith("_Xs"):
            continue
        Xs, want = d[k], d[k[:-3] + "_actions"]
        B, N, D, K1 = Xs.shape
        model = mrsgym_amd.Reynolds(N=N, D=D, K=K1 - 1, OUT_DIM=3)
        got = model.forward(None, torch.from_numpy(Xs)).cpu().numpy()
        assert np.array_equal(got, oracle.reynolds(Xs[..., 1])), k
        assert np.abs(got - want).max() < 2e-6, k
        stack = torch.from_numpy(np.ascontiguousarray(Xs.transpose(0, 3, 1, 2)))       # (B, K+1, N, D)
        assert np.array_equal(model.forward_stack(stack).cpu().numpy(), got)
    # K > 1: the reference's controller reads hops 0 and 1 only (Reynolds_Node.py:30): covered by the K2 / K3 fixtures above
    with pytest.raises(NotImplementedError):
        mrsgym_amd.Reynolds(N=4, D=6, K=0)
    # closed loop: the expert drives a small flock through the Gym surface, reading X(t-1) from the ring in place
    E, N = 3, 12
    pos, eul = grid_spawn(E, N, seed=2)
    env = mrsgym_amd.make('mrs-v0', N_ENVS=E, N_AGENTS=N, state_fn=state_fn, K_HOPS=1, COMM_RANGE=2.5,
                          START_POS=torch.from_numpy(pos), ACTION_TYPE="set_target_vel")
    model = mrsgym_amd.Reynolds(N=N, D=6)
    X = env.get_Xk()
    for t in range(30):
        a = model.from_env(env)
        assert np.array_equal(a.cpu().numpy(), oracle.reynolds(X[:, 1].cpu().numpy()))
        assert float(a.norm(dim=-1).max()) <= 1.0 + 1e-6
        X, r, done, info = env.step(a)
    assert torch.isfinite(X).all()


def test_flocking_metrics_match_the_reference_analytics(golden_dir):
    """mrs_flock_metrics / mrsgym_amd.MRSAnalytics against the reference's own MRSAnalytics outputs (tests/golden/F8,
    generated from examples/simulating_data/helper/MRSAnalytics.py:13-101).  separation and cohesion are float32
    norms + min/max: bit-exact; the means (torch's own summation order) and the determinant to 1e-5 relative."""
    import mrsgym_amd
    d = np.load(os.path.join(golden_dir, "F8_flock_metrics.npz"))
    n = 0
    for key in ("N3", "N12", "N64"):
        X = torch.from_numpy(d[key + "_X"]).cuda()
        an = mrsgym_amd.MRSAnalytics(X)
        assert (an.num_episodes, an.episode_length, an.N) == tuple(X.shape[:3])
        for name in ("separation", "cohesion"):
            got = getattr(an, name)().cpu().numpy()
            assert got.shape == d[key + "_" + name].shape
            assert np.array_equal(got, d[key + "_" + name]), (key, name, np.abs(got - d[key + "_" + name]).max())
        assert np.array_equal(an.cohesion(exclude_leader=True).cpu().numpy(), d[key + "_cohesion_noleader"])
        for name, tol in (("dist_to_leader", 1e-6), ("vel_stddev", 1e-5), ("vel_mag", 1e-6), ("vel_leader_alignment", 1e-6)):
            if name == "vel_stddev" and key == "N3":
                continue    # three velocities minus their mean span a plane: the determinant is 0 up to rounding noise (the reference's float32 LU returns 1e-5 or NaN)
            got = getattr(an, name)().cpu().numpy()
            np.testing.assert_allclose(got, d[key + "_" + name], rtol=tol, atol=tol, err_msg=key + " " + name)
        for name in ("separation_avg", "cohesion_avg", "vel_stddev_avg", "vel_mag_avg", "vel_leader_alignment_avg"):
            if (name == "vel_stddev_avg" and key == "N3") or not np.isfinite(d[key + "_" + name]):
                continue
            assert abs(float(getattr(an, name)()) - float(d[key + "_" + name])) < 1e-5 * max(1.0, abs(float(d[key + "_" + name])))
        n += 1
    assert n == 3
    # NaN-padded short episodes (Trainer.get_episodes) propagate like torch: that frame's metrics are NaN, the others untouched
    X = torch.from_numpy(d["N12_X"]).cuda().clone()
    ref = mrsgym_amd.MRSAnalytics(X.clone())
    X[1, -2:] = float("nan")
    an = mrsgym_amd.MRSAnalytics(X)
    assert torch.isnan(an.cohesion()[1, -2:]).all() and torch.isnan(an.separation()[1, -2:]).all()
    assert torch.equal(an.cohesion()[0], ref.cohesion()[0]) and torch.equal(an.separation()[1, :-2], ref.separation()[1, :-2])
    # straight from a rollout log of the vectorised data generator
    log = mrsgym_amd.RolloutLog(2, 12, 6, capacity=7)
    for t in range(7):
        log.set_state(torch.zeros(2, 12, 1, dtype=torch.int64, device="cuda"), X[:2, t], done=torch.tensor([t == 3, False]))
    ep = log.get_episodes()
    assert ep["X"].shape == (3, 7, 12, 6)                     # env 0: 4 + 3 steps, env 1: one episode of 7
    an = mrsgym_amd.MRSAnalytics(log)
    assert an.separation().shape == (3, 7, 12) and torch.isnan(an.cohesion()[0, 4:]).all() and torch.isfinite(an.cohesion()[2, :5]).all()


def test_data_generation_loop_like_the_reference_example():
    """The reference's data generator (examples/simulating_data/gen_data.py + helper/DataGenerator.py:8-47), vectorised:
    Reynolds expert -> leader override -> log sample -> env.step, episodes end when an agent loses all neighbours
    (gen_data.py:57-61) or after episode_length steps, finished envs restart on their own (AUTO_RESET)."""
    import mrsgym_amd
    from mrsgym_amd.util import CombinedDistribution
    from torch.distributions import Normal, Uniform
    E, N, STEPS, EP_LEN = 16, 12, 60, 25
    dist = CombinedDistribution([Normal(torch.zeros(N, 2), 1.25), Uniform(2.0 * torch.ones(N, 1), 5.0 * torch.ones(N, 1))],
                                mixer='cat', dim=1)                              # gen_data.py:27-29

    def done_fn(A=None, steps_since_reset=0, env=None, **kw):                     # gen_data.py:57-61, per env
        isolated = (A[:, 0].sum(dim=-1) == 0).any(dim=-1)
        return isolated | (mrs.env_steps() + 1 >= EP_LEN)
    mrs = mrsgym_amd.make('mrs-v0', N_ENVS=E, N_AGENTS=N, state_fn=state_fn, K_HOPS=1, COMM_RANGE=2.5, START_POS=dist,
                          ACTION_TYPE='set_target_vel', done_fn=done_fn, AUTO_RESET=True, SEED=3)
    model = mrsgym_amd.Reynolds(N=N, D=6, K=1, OUT_DIM=3)
    log = mrsgym_amd.RolloutLog(E, N, 6, capacity=STEPS)
    X = mrs.get_Xk()
    A = mrs.calc_Ak()                                                           # DataGenerator.py:22
    resets = 0
    for t in range(STEPS):
        action = model.from_env(mrs)
        expert = action.clone()
        action[:, 0, :] = torch.tensor([0.3, 0.0, 0.0], device=action.device)     # the leader's own velocity (gen_data.py:76-88)
        Xn, r, done, info = mrs.step(action)
        log.set_state(A[:, 0], X[:, 0], done=done, expert=expert)
        resets += int(done.sum())
        X, A = Xn, info["A"]
    assert len(log) == STEPS and resets >= E * (STEPS // EP_LEN)                  # every env restarted at least twice
    d = log.trainer_dict()["history"]
    assert len(d["X"]) == E * STEPS and sum(d["done"]) >= resets
    ex = torch.stack(list(d["expert"]))
    assert torch.isfinite(ex).all() and float(ex.norm(dim=-1).max()) <= 1.0 + 1e-6
    assert torch.isfinite(torch.stack(list(d["X"]))).all()


def test_bench_contract_single_and_two_ranks():
    """bench.py end to end at a small size: one JSON line with the contract's keys for N=1, and the N>1 launch the
    driver uses (torch.distributed.run, one rank per process) rehearsed with two gloo ranks sharing this one GPU:
    weak scaling, `value` with the per-step joint-observation all-gather, the exchange-free rate beside it."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MRS_BENCH_PREWARM_S="0")
    common = ["--steps", "20", "--warmup", "5", "--envs-per-gpu", "64"]
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline"] + common, env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["value"] > 0 and d["unit"] == "agent-steps/s"
    assert d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1 and d["config"]["workload"]
    assert d["rollin_steps"] == 700 and d["dense_a"]["value"] > 0 and d["dense_a"]["ms_per_step"] > 0
    # the fidelity knobs `value` was measured with (the library's defaults) and the two legs beside it
    m = d["config"]["fidelity"]
    assert (m["solver_iters"], m["round_euler_readback"], m["rest_shortcut"], m["pair_contact"]) == (10, 0, 1, 1)
    assert d["literal"]["fidelity"]["round_euler_readback"] == 1 and d["literal"]["fidelity"]["rest_shortcut"] == 0 and d["literal"]["fidelity"]["solver_iters"] == 10
    assert d["solver6"]["fidelity"]["solver_iters"] == 6 and d["literal"]["kernel_ms"] > 0 and 0 < d["solver6"]["roofline_frac"] < 1
    env2 = dict(env, MRS_BENCH_SINGLE_DEVICE="1", MRS_DIST_BACKEND="gloo", MRS_BENCH_DIRECT="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29533", os.path.join(root, "bench.py"),
                          "--gpus", "2"] + common, env=env2, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1                                       # rank 0 only
    d2 = json.loads(lines[0])
    assert d2["n_gpus"] == 2 and d2["scaling"] == "weak" and d2["value"] > 0
    # N > 1: `value` includes the per-step joint-observation all-gather (BASELINE config 5); the exchange-free rate is beside it
    assert "all-gather" in d2["config"]["parallelism"]
    assert d2["no_exchange"]["value"] > 0 and d2["obs_allgather"]["bytes_sent_per_rank_per_step"] == 64 * 64 * 6 * 4
    assert d2["obs_allgather"]["xgmi_floor_ms"] > 0
    # gathered every k-th step only: first-class legs; the one-shot form needs RCCL (one device per rank) and says so here
    assert d2["gather_every_4"]["value"] > 0 and d2["gather_every_16"]["value"] > 0
    assert d2["direct_p2p"]["value"] is None and "nccl" in d2["direct_p2p"]["error"]
    # what the collective layer saw, from the JSON line alone: here two gloo ranks rehearsing on ONE device
    c = d2["comm"]
    assert c["backend"] == "gloo" and c["world_size"] == 2 and len(c["rank_devices"]) == 2 and [r["rank"] for r in c["ranks"]] == [0, 1]
    assert c["distinct_devices"] == 1 and c["single_device_rehearsal"] is True and c["rccl_version"]
    assert "comm" not in d


def test_handle_on_a_device_that_is_not_the_current_one():
    """a handle is bound to its device; every launching entry point selects it for the call (DeviceGuard in
    mrs_kernels.hip) whatever torch's current device is.  Needs two GPUs: skipped on the one-GPU box."""
    import torch, mrsgym_amd
    if torch.cuda.device_count() < 2 or os.environ.get("MRS_TEST_MULTI_DEVICE") != "1":
        pytest.skip("needs two visible GPUs and MRS_TEST_MULTI_DEVICE=1 (never run on the one-GPU boxes this build had)")
    torch.cuda.set_device(0)
    E, N = 4, 12
    pos, eul = grid_spawn(E, N)
    z = np.zeros((E, N, 3), np.float32)
    out = []
    for dev in ("cuda:0", "cuda:1"):
        sh = mrsgym_amd.SwarmShard(E, N, dev)
        sh.set_state(pos=pos, ori=eul, vel=z, angvel=z)
        a = torch.from_numpy(ActionStream("set_target_vel", E, N, pos, seed=3)(0)).to(dev)
        obs = torch.zeros(E, N, sh.D, device=dev)
        for _ in range(5):
            with torch.cuda.device(0):        # the current device stays 0 throughout
                sh.step(a, "set_target_vel", obs_out=obs)
        torch.cuda.synchronize(dev)
        out.append(obs.cpu().numpy())
    np.testing.assert_array_equal(out[0], out[1])


def test_round_euler_readback_kwarg_reaches_the_kernel():
    """MRS(..., ROUND_EULER_READBACK=True) selects the attitude controller's literal float32 rounding of the Euler
    read-back (include/mrs_hip.h; docs/experiments.md section 4 deviation 7): the flag arrives in the shard's parameters, and one
    step of the two forms differs by what that rounding is worth -- more than nothing, less than 1e-6 rad/s."""
    import mrsgym_amd
    E, N = 3, 12
    pos, eul = grid_spawn(E, N, seed=2, yaw_range=0.8)
    a = torch.from_numpy(ActionStream("set_target_vel", E, N, pos, seed=9, coherent=True)(0)).cuda()
    w = {}
    for flag in (False, True):
        env = mrsgym_amd.make('mrs-v0', N_ENVS=E, N_AGENTS=N, state_fn=state_fn, START_POS=torch.from_numpy(pos),
                              ACTION_TYPE="set_target_vel", ROUND_EULER_READBACK=flag)
        assert int(env.shard.params.round_euler_readback) == int(flag)
        env.reset(ori=torch.from_numpy(eul))
        env.step(a)
        w[flag] = env.shard.view(env.shard.angvel).clone()
    d = float((w[True] - w[False]).abs().max())
    assert 0.0 < d < 1e-6, d


def test_quad_contact_kwarg_reaches_the_kernel():
    """MRS(..., QUAD_CONTACT=...) -> MrsParams.pair_contact: two quadcopters 0.13 m apart (inside the
    contact range 2 r + threshold = 0.14) closing at 3 m/s lose, in one step, the part of the closing velocity that would
    take them past touching (gap / dt = 1 m/s stays) -- 1 m/s each -- with it, and nothing without."""
    import mrsgym_amd
    pos = torch.tensor([[[0.0, 0.0, 20.0], [0.13, 0.0, 20.0]]])
    vx = {}
    for flag in (True, False):
        env = mrsgym_amd.make('mrs-v0', N_ENVS=1, N_AGENTS=2, state_fn=state_fn, START_POS=pos, ACTION_TYPE="set_target_vel",
                              QUAD_CONTACT=flag)
        assert int(env.shard.params.pair_contact) == int(flag)
        env.reset(vel=torch.tensor([[[1.5, 0.0, 0.0], [-1.5, 0.0, 0.0]]]))
        env.step(torch.tensor([[[1.5, 0.0, 0.0], [-1.5, 0.0, 0.0]]]).cuda())
        vx[flag] = env.shard.view(env.shard.vel)[0, :, 0].cpu().numpy()
    assert abs(vx[False][0] - 1.5) < 0.05 and abs(vx[False][1] + 1.5) < 0.05, vx
    assert abs(vx[True][0] - 0.5) < 0.05 and abs(vx[True][1] + 0.5) < 0.05, vx


def test_solver_iters_and_rest_shortcut_kwargs_reach_the_kernel():
    """MRS(..., SOLVER_ITERS=k, REST_SHORTCUT=flag) -> MrsParams.solver_iters / rest_shortcut (include/mrs_hip.h).  Bodies dropped
    flat onto the ground: with the shortcut they are finished in their own lane once they lie still, without it they keep going
    through the sweeps -- the two agree to the shortcut's stated 1e-5 m/s per step, and a cap of 1 sweep
    leaves a tilted touchdown visibly less converged than the default 10."""
    import mrsgym_amd
    E, N = 2, 12
    pos, eul = grid_spawn(E, N, seed=3)
    pos[..., 2] = 0.53
    flat = np.zeros_like(eul)
    out = {}
    for key, kw, ori in (("on", {}, flat), ("off", dict(REST_SHORTCUT=False), flat), ("k1", dict(SOLVER_ITERS=1), eul * 0 + [0.3, -0.2, 0.5]),
                         ("k10", {}, eul * 0 + [0.3, -0.2, 0.5])):
        env = mrsgym_amd.make('mrs-v0', N_ENVS=E, N_AGENTS=N, state_fn=state_fn, START_POS=torch.from_numpy(pos), ACTION_TYPE="set_target_vel", **kw)
        prm = env.shard.params
        assert int(prm.rest_shortcut) == int(kw.get("REST_SHORTCUT", True)) and int(prm.solver_iters) == int(kw.get("SOLVER_ITERS", 10))
        env.reset(ori=torch.from_numpy(np.ascontiguousarray(ori, dtype=np.float32)))
        for _ in range(120):
            env.step(None)                       # no rotor forces: the bodies fall 1.75 cm and come to rest
        sh = env.shard
        out[key] = torch.cat([sh.view(sh.pos), sh.view(sh.vel), sh.view(sh.angvel)], -1).cpu().numpy()
    assert np.abs(out["on"][..., 2] - 0.5125).max() < 1e-4            # lying on the ground top, hull half-length above it
    assert np.abs(out["on"] - out["off"]).max() < 1e-4 and np.abs(out["on"][..., 3:]).max() < 1e-4
    assert np.abs(out["k1"] - out["k10"]).max() > 1e-6                 # the cap does reach the sweeps
    with pytest.raises(ValueError):
        mrsgym_amd.make('mrs-v0', N_AGENTS=3, state_fn=state_fn, SOLVER_ITERS=0)
