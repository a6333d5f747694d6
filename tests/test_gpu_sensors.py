"""the reference's geometry sensors (Object.py:100-174) as batched kernels (mrs_raycast,
mrs_proximity through the C-ABI) against the oracle's restatement (oracle/mrs_sensors.c, brute-force certified in
tests/test_oracle_sensors.py), and through the QuadView / Environment surface the reference's examples use.
Parity with pybullet's own answers is unpinned (pybullet absent; the reference holds no fixtures)."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation as R

import oracle

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _scene(rng, E, N, spread=0.3, z=(0.6, 1.4)):
    pos = np.concatenate([rng.uniform(-spread, spread, (E, N, 2)), rng.uniform(z[0], z[1], (E, N, 1))], -1)
    quat = R.random(E * N, random_state=int(rng.integers(1 << 30))).as_quat().reshape(E, N, 4)
    return pos, quat


def _shard(pos, quat):
    import mrsgym_amd
    E, N = pos.shape[:2]
    sh = mrsgym_amd.SwarmShard(E, N, "cuda:0")
    sh.set_state_f64(pos=pos, quat=quat, vel=np.zeros((E, N, 3)), angvel=np.zeros((E, N, 3)))
    return sh


@pytest.mark.parametrize("E,N", [(3, 5), (2, 64), (1, 1), (2, 130), (1, 1024)])   # N = 1024: 96 KB of LDS, above the launch's default limit
def test_raycast_matches_oracle(E, N):
    rng = np.random.default_rng(E * 100 + N)
    pos, quat = _scene(rng, E, N, spread=0.3 if N < 20 else 1.5)
    sh = _shard(pos, quat)
    dirs = rng.normal(size=(9, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    dirs[0] = [0, 0, -1]
    off = np.array([0, 0, -0.1], np.float32)
    hits = 0
    for body, RANGE in ((True, 2.0), (False, 100.0)):
        out = sh.raycast(off, dirs, body=body, RANGE=RANGE)
        obj, pw, pb, dist = (out[k].cpu().numpy() for k in ("object", "pos world", "pos", "dist"))
        assert obj.shape == (E, N, 9) and pw.shape == (E, N, 9, 3) and dist.shape == (E, N, 9)
        for e in range(E):
            for i in range(0, N, max(1, N // 7)):
                want = oracle.raycast(pos[e], quat[e], i, off, dirs.copy(), body=body, RANGE=RANGE)
                # a ray grazing an edge may be a hit in one float64 evaluation order and a miss in the other
                same = obj[e, i] == want["object"]
                assert same.mean() >= 0.85, (e, i, obj[e, i], want["object"])
                tol = 1e-5 * max(1.0, RANGE / 10)
                np.testing.assert_allclose(pw[e, i][same], want["pos world"][same], atol=tol)
                np.testing.assert_allclose(pb[e, i][same], want["pos"][same], atol=tol)
                np.testing.assert_allclose(dist[e, i][same], want["dist"][same], atol=tol)
                hits += int((want["object"][same] >= 0).sum())
    assert hits > 10
    # the downward ray of a quadcopter above the ground box always finds the ground or a neighbour below it
    assert (out["object"][..., 0] >= 0).all()


@pytest.mark.parametrize("centre", [(0.0, 0.0), (13.0, -13.0), (-14.2, 14.2)])
def test_raycast_aimed_rays_survive_the_float32_cull(centre):
    """Round 5: the bounding-sphere cull in front of the exact ray / cylinder test runs in float32 with margins (mrs_sensors.hpp).
    Rays aimed from agent 0 at the centre of every other agent -- and the same rays shifted sideways by most of the collision
    radius -- cannot be lost to it wherever the swarm sits on the 30 m ground box: hit object and distance as the oracle has them."""
    rng = np.random.default_rng(int(abs(centre[0]) * 10) + 3)
    E, N = 12, 9
    pos, quat = _scene(rng, E, N, spread=1.2, z=(0.8, 2.5))
    pos[..., 0] += centre[0]; pos[..., 1] += centre[1]
    sh = _shard(pos, quat)
    checked = 0
    for e in range(E):
        d = pos[e, 1:] - pos[e, 0]
        dist = np.linalg.norm(d, axis=1)
        dirs = (d / dist[:, None]).astype(np.float32)
        for shift in (0.0, 0.045):
            side = np.cross(dirs, [0, 0, 1.0]); side /= np.maximum(np.linalg.norm(side, axis=1, keepdims=True), 1e-9)
            off = (shift * side[0]).astype(np.float32)                          # one offset per call (Object.raycast: offset is shared)
            out = sh.raycast(off, dirs, body=False, RANGE=float(dist.max() + 1.0))
            obj = out["object"][e, 0].cpu().numpy(); dd = out["dist"][e, 0].cpu().numpy()
            want = oracle.raycast(pos[e], quat[e], 0, off, dirs.copy(), body=False, RANGE=float(dist.max() + 1.0))
            if shift == 0.0:
                assert (want["object"] >= 0).all() and (want["object"] != 0).all()   # aimed at a centre: something is hit on the way
            same = obj == want["object"]
            assert same.all(), (e, shift, obj, want["object"])
            np.testing.assert_allclose(dd, want["dist"], atol=2e-5)
            checked += len(obj)
    assert checked == E * 2 * (N - 1)


@pytest.mark.parametrize("E,N", [(4, 6), (2, 64), (1, 1)])
def test_proximity_matches_oracle(E, N):
    rng = np.random.default_rng(7 + N)
    pos, quat = _scene(rng, E, N, spread=0.12 if N < 20 else 0.8, z=(0.52, 0.9))
    sh = _shard(pos, quat)
    dist, ps, po = sh.proximity(points=True)
    d = dist.cpu().numpy()
    assert d.shape == (E, N, N + 1)
    for e in range(E):
        want = oracle.proximity(pos[e], quat[e])
        np.testing.assert_allclose(d[e], want, atol=2e-6)          # float32 outputs of a float64 GJK (1e-8) / closed form
        for i in range(0, N, max(1, N // 5)):
            o = oracle.closest(pos[e], quat[e], i)
            far = want[i] > 1e-4                                   # touching / overlapping pairs: any common point will do
            # the closest PAIR is unique only up to the flat faces; its length is what is compared
            seg = np.linalg.norm(ps[e, i].cpu().numpy() - po[e, i].cpu().numpy(), axis=-1)
            np.testing.assert_allclose(seg[far], np.abs(want[i])[far], atol=5e-6)
            np.testing.assert_allclose(ps[e, i, N].cpu().numpy(), o["closest pos self"][N], atol=1e-6)   # the ground pair is unique
    if N > 1:
        assert (d[:, :, :N][:, np.arange(N), np.arange(N)] == 0).all()
        assert np.allclose(d[:, :, :N], np.swapaxes(d[:, :, :N], 1, 2), atol=2e-6)
        # MAX_DIST semantics: pairs beyond it are reported as +inf (the reference returns an empty result), never wrongly
        cut = float(np.median(d[:, :, :N]))
        dc = sh.proximity(max_dist=cut).cpu().numpy()[:, :, :N]
        assert np.all(dc[np.isfinite(dc)] == d[:, :, :N][np.isfinite(dc)]) and np.all(d[:, :, :N][~np.isfinite(dc)] > cut)


def state_fn(quad):
    return torch.cat([quad.get_pos(), quad.get_vel()])


def test_object_level_sensors_like_the_reference_examples():
    """examples/object_functions/{raycast,closest_points,closest_objects}.py and the collision reward of
    examples/env_examples/magent.py:38-43, on one env and vectorised."""
    import mrsgym_amd
    N = 4
    pos = torch.tensor([[0., 0., 1.0], [0.5, 0., 0.9], [0.0, 0.11, 1.0], [3.0, 3.0, 0.5125]])   # 2 touches 0; 3 rests on the ground
    env = mrsgym_amd.make('mrs-v0', N_AGENTS=N, state_fn=state_fn, START_POS=pos, ACTION_TYPE='set_target_pos')
    env.reset(ori=torch.zeros(N, 3))
    q0, q1, q2, q3 = env.get_agents()
    # raycast.py: four rays from 10 cm below the body
    rays = q0.raycast(offset=torch.tensor([0., 0., -0.1]), directions=torch.tensor([[1., 0., 0.], [0., 1., 0.], [0., -1., 0.], [0., 0., -1.]]), body=True)
    assert rays["object"][0] is q1 and rays["object"][1] is None and rays["object"][3] is env.get_env().ground
    np.testing.assert_allclose(rays["pos"][0].cpu().numpy(), [0.44, 0, 0], atol=1e-6)
    np.testing.assert_allclose(rays["pos"][3].cpu().numpy(), [0, 0, -0.4], atol=1e-6)
    np.testing.assert_allclose(rays["dist"].cpu().numpy(), [0.44, 0, 0, 0.4], atol=1e-6)
    assert rays["pos"].reshape(12).shape == (12,)                        # what the example's state_fn returns
    # closest_points.py
    d = q0.get_dist(q1)
    # rim to rim: 0.5 - 2 rc apart horizontally, 0.1 - 2 hl vertically
    assert d["distance"].shape == (1,) and abs(float(d["distance"]) - np.hypot(0.5 - 0.12, 0.1 - 0.025)) < 1e-6
    assert q0.get_dist(q1, MAX_DIST=0.1)["distance"].shape == (0,)
    c = q0.get_contact_points()
    assert c["object"] == [q2] and c["pos"].shape == (1, 3) and float(c["distance"][0]) <= 0.02
    assert q0.get_contact_points(q1)["object"] == [] and q3.get_contact_points()["object"] == [env.get_env().ground]
    # magent.py:42  reward = -1 if agent.collision() else 0
    assert [a.collision() for a in env.get_agents()] == [True, False, True, True]
    assert env.get_env().collisions().cpu().tolist() == [True, False, True, True]
    # closest_objects.py
    near = q0.get_closest_objects(radius=0.45)
    assert q0 in near and q1 in near and q2 in near and q3 not in near and env.get_env().ground not in near
    assert env.get_env().ground in q0.get_closest_objects(radius=0.6)
    # a raycasting state_fn (raycast.py:14-19) runs as the env's observation
    def ray_state(quad):
        r = quad.raycast(offset=torch.tensor([0., 0., -0.1]), directions=torch.tensor([[1., 0., 0.], [0., 0., -1.]]), body=True)
        return r["pos"].reshape(6)
    with pytest.warns(RuntimeWarning):
        env2 = mrsgym_amd.make('mrs-v0', N_AGENTS=N, state_fn=ray_state, START_POS=pos)
    X = env2.reset(ori=torch.zeros(N, 3))
    assert X.shape == (1, N, 6) and abs(float(X[0, 0, 0]) - 0.44) < 1e-6 and abs(float(X[0, 0, 5]) + 0.4) < 1e-6


def test_vectorised_collision_reward_over_many_envs():
    """The magent-style collision reward for E x N quadcopters in one launch, while the swarm drops onto the ground."""
    import mrsgym_amd
    from util_scenarios import grid_spawn
    E, N = 64, 16
    pos, eul = grid_spawn(E, N)
    pos[..., 2] = 0.6 + 0.2 * (pos[..., 2] - 1.0)

    def reward_fn(env=None, **kw):
        return -env.collisions().float()                                  # (E,N)
    env = mrsgym_amd.make('mrs-v0', N_ENVS=E, N_AGENTS=N, state_fn=state_fn, START_POS=torch.from_numpy(pos), reward_fn=reward_fn,
                          ACTION_TYPE='set_target_vel')
    env.reset(ori=torch.from_numpy(eul))
    total = 0.0
    for t in range(120):                                                   # motors off (MRS.py:243-253): the swarm drops and comes to rest
        X, r, d, info = env.step(None)
        assert r.shape == (E, N)
        total += float(r.sum())
    z = X[:, 0, :, 2]
    on_ground = (z < 0.53)
    assert on_ground.float().mean() > 0.5                                 # most of them landed ...
    assert torch.equal(env.get_env().collisions(ground=True) | ~on_ground, torch.ones_like(on_ground))   # ... and report it
    assert total < 0


def test_batched_contact_points_and_closest_objects_match_the_one_env_forms():
    """Object.get_contact_points / get_closest_objects (Object.py:100-116, :140-147) with N_ENVS > 1: padded tensors + counts,
    env by env the content of the one-env (list) forms."""
    import mrsgym_amd
    E, N = 5, 6
    rng = np.random.default_rng(3)
    pos = np.concatenate([rng.uniform(-0.15, 0.15, (E, N, 2)), rng.uniform(0.51, 0.62, (E, N, 1))], -1).astype(np.float32)
    pos[:, 0] = [0, 0, 0.5125]                                            # agent 0 rests on the ground in every env
    pos[:, 1, :2] = [[0.05 + 0.02 * e, 0.0] for e in range(E)]            # agent 1 next to it
    env = mrsgym_amd.make('mrs-v0', N_ENVS=E, N_AGENTS=N, state_fn=state_fn, START_POS=torch.from_numpy(pos))
    env.reset(ori=torch.zeros(E, N, 3))
    singles = []
    for e in range(E):
        one = mrsgym_amd.make('mrs-v0', N_AGENTS=N, state_fn=state_fn, START_POS=torch.from_numpy(pos[e]))
        one.reset(ori=torch.zeros(N, 3))
        singles.append(one)
    for i in (0, 1, 3):
        q = env.get_agents()[i]
        for other in (None, 1 if i != 1 else 0):
            c = q.get_contact_points(other=other)
            assert c["object"].shape == (E, N) and c["pos"].shape == (E, N, 3) and c["distance"].shape == (E, N) and c["count"].shape == (E,)
            for e in range(E):
                s = singles[e].get_agents()[i].get_contact_points(other=None if other is None else singles[e].get_agents()[other])
                n = int(c["count"][e])
                assert n == len(s["object"])
                want = [o.uid for o in s["object"]]
                assert c["object"][e, :n].tolist() == want and bool((c["object"][e, n:] == -1).all())
                if n:
                    np.testing.assert_allclose(c["pos"][e, :n].cpu().numpy(), s["pos"].cpu().numpy(), atol=1e-6)
                    np.testing.assert_allclose(c["distance"][e, :n].cpu().numpy(), s["distance"].cpu().numpy(), atol=1e-6)
        for radius in (0.1, 0.3):
            g = q.get_closest_objects(radius)
            for e in range(E):
                want = [o.uid for o in singles[e].get_agents()[i].get_closest_objects(radius)]
                n = int(g["count"][e])
                assert g["object"][e, :n].tolist() == want and bool((g["object"][e, n:] == -1).all())
    assert int(env.get_agents()[0].get_contact_points()["count"].min()) >= 1     # the ground, at least
