"""The oracle's geometry sensors (oracle/mrs_sensors.c: Object.py:100-174 against the analytic scene) checked by
brute force: closest points certified through the separating-plane condition of convex sets, ray hits against a
dense march along the segment.  CPU only.  (Parity with pybullet itself is unpinned: see mrs_sensors.c.)"""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation as R

import oracle

RC, HL, GZ = 0.06, 0.0125, 0.5


def _scene(rng, n, spread=0.5, z=(0.6, 1.6)):
    pos = np.concatenate([rng.uniform(-spread, spread, (n, 2)), rng.uniform(z[0], z[1], (n, 1))], 1)
    quat = R.random(n, random_state=int(rng.integers(1 << 30))).as_quat()
    return pos, quat


def _axis(q):
    return R.from_quat(q).as_matrix()[:, 2]


def _support(c, a, d):
    """farthest point of the cylinder (centre c, axis a) in direction d"""
    da = d @ a
    rad = d - da * a
    n = np.linalg.norm(rad)
    p = c + np.sign(da if da != 0 else 1.0) * HL * a
    return p + (RC / n) * rad if n > 1e-12 else p


def _inside(c, a, x, tol=1e-9):
    y = x - c
    t = y @ a
    return abs(t) <= HL + tol and np.linalg.norm(y - t * a) <= RC + tol


def test_cylinder_cylinder_closest_points_are_optimal():
    rng = np.random.default_rng(1)
    n_checked = 0
    for trial in range(60):
        n = 6
        pos, quat = _scene(rng, n, spread=0.15 if trial % 2 else 0.6, z=(0.9, 1.2) if trial % 2 else (0.6, 1.6))
        for i in range(n):
            out = oracle.closest(pos, quat, i)
            for j in range(n):
                if j == i:
                    continue
                d, pa, pb = out["distance"][j], out["closest pos self"][j], out["closest pos other"][j]
                ai, aj = _axis(quat[i]), _axis(quat[j])
                assert _inside(pos[i], ai, pa, 1e-7) and _inside(pos[j], aj, pb, 1e-7)   # sliver simplices on the curved rim: 1e-8
                if d == 0:      # overlap: some point belongs to both
                    assert np.allclose(pa, pb) and _inside(pos[j], aj, pa, 1e-6) and _inside(pos[i], ai, pa, 1e-6)
                    continue
                assert abs(np.linalg.norm(pa - pb) - d) < 1e-12
                nrm = (pa - pb) / d
                # no point of A lies further along -n than pa, none of B further along +n than pb: the slab between
                # the two tangent planes separates the sets, so d is the distance
                assert (-nrm) @ _support(pos[i], ai, -nrm) <= (-nrm) @ pa + 1e-7      # (GJK stops at 1e-14 relative on d^2: 1e-7 on the direction)
                assert nrm @ _support(pos[j], aj, nrm) <= nrm @ pb + 1e-7
                n_checked += 1
    assert n_checked > 1000
    # symmetric, and the all-pairs form agrees
    pos, quat = _scene(rng, 8)
    D = oracle.proximity(pos, quat)
    assert np.allclose(D[:, :8], D[:, :8].T, atol=1e-12) and np.all(np.diag(D[:, :8]) == 0)
    for i in range(8):
        assert np.allclose(D[i], oracle.closest(pos, quat, i)["distance"], atol=1e-12)


def test_known_configurations():
    q0 = np.array([0, 0, 0, 1.0])
    pos = np.array([[0, 0, 1.0], [0.3, 0, 1.0], [0, 0, 2.0], [0.05, 0, 1.02]])
    quat = np.tile(q0, (4, 1))
    out = oracle.closest(pos, quat, 0)
    np.testing.assert_allclose(out["distance"], [0, 0.3 - 2 * RC, 1 - 2 * HL, 0, 1 - HL - GZ], atol=1e-12)   # side by side, stacked, overlapping, ground
    # on edge: axis horizontal -> lowest point is the rim, rc below the centre
    qe = R.from_euler("x", 90, degrees=True).as_quat()
    out = oracle.closest(np.array([[0, 0, 1.0]]), qe[None], 0)
    assert abs(out["distance"][1] - (1 - RC - GZ)) < 1e-12
    np.testing.assert_allclose(out["closest pos self"][1], [0, 0, 1 - RC], atol=1e-12)
    np.testing.assert_allclose(out["closest pos other"][1], [0, 0, GZ], atol=1e-12)
    # tilted by 30 degrees about x: lowest point = lower rim, hl cos + rc sin below the centre
    t = np.radians(30)
    qt = R.from_euler("x", 30, degrees=True).as_quat()
    out = oracle.closest(np.array([[0, 0, 1.0]]), qt[None], 0)
    assert abs(out["distance"][1] - (1 - (HL * np.cos(t) + RC * np.sin(t)) - GZ)) < 1e-12
    # resting on the ground, slightly sunk: signed (negative) distance
    out = oracle.closest(np.array([[0, 0, GZ + HL - 0.001]]), q0[None], 0)
    assert abs(out["distance"][1] + 0.001) < 1e-12


def _march(pos, quat, o, d, steps=40001):
    """first parameter t in [0,1] at which o + t d is inside the ground box or a cylinder it did not start in"""
    t = np.linspace(0, 1, steps)
    x = o[None] + t[:, None] * d[None]
    hit_t, hit_obj = None, -1
    inside_g = (np.abs(x[:, 0]) <= 15) & (np.abs(x[:, 1]) <= 15) & (x[:, 2] <= GZ) & (x[:, 2] >= GZ - 1)
    cands = []
    if not inside_g[0] and inside_g.any():
        cands.append((t[np.argmax(inside_g)], len(pos)))
    for j in range(len(pos)):
        a = _axis(quat[j])
        y = x - pos[j]
        ax = y @ a
        ins = (np.abs(ax) <= HL) & (np.linalg.norm(y - ax[:, None] * a[None], axis=1) <= RC)
        if not ins[0] and ins.any():
            cands.append((t[np.argmax(ins)], j))
    if cands:
        hit_t, hit_obj = min(cands)
    return hit_t, hit_obj


def test_raycast_against_a_dense_march():
    rng = np.random.default_rng(3)
    hits = 0
    for trial in range(25):
        n = 5
        pos, quat = _scene(rng, n, spread=0.25, z=(0.7, 1.3))
        agent = int(rng.integers(n))
        dirs = rng.normal(size=(12, 3)).astype(np.float32)
        dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
        off = np.array([0, 0, -0.1], np.float32)
        for body in (True, False):
            RANGE = 2.0
            out = oracle.raycast(pos, quat, agent, off, dirs.copy(), body=body, RANGE=RANGE)
            ob = oracle.observe(pos[agent], quat[agent], np.zeros(3), np.zeros(3))
            Rm, p32 = ob["mat"].astype(np.float32), ob["pos"]
            for r in range(len(dirs)):
                ofw = Rm @ off if body else off
                dw = Rm @ (dirs[r] * np.float32(RANGE)) if body else dirs[r] * np.float32(RANGE)
                st = (ofw + p32).astype(np.float64)
                t, obj = _march(pos, quat, st, dw.astype(np.float64))
                if obj < 0:
                    assert out["object"][r] == -1 and np.all(out["pos"][r] == 0) and out["dist"][r] == 0
                    continue
                # the march may miss a graze thinner than its step; the oracle's hit must then be confirmed by it otherwise
                if out["object"][r] != obj:
                    pytest.fail("object mismatch: oracle %d march %d (t=%g)" % (out["object"][r], obj, t))
                hits += 1
                # Object.py:171: dist = |body-frame pos| = distance from the ray start to the hit
                assert abs(out["dist"][r] - t * np.linalg.norm(dw)) < 2 * np.linalg.norm(dw) / 40000 + 1e-5
                np.testing.assert_allclose(out["pos world"][r] + ofw, st + t * dw, atol=2e-4)
                np.testing.assert_allclose(Rm @ out["pos"][r], out["pos world"][r] - p32, atol=1e-5)
    assert hits > 100


def test_raycast_reference_example_layout():
    """examples/object_functions/raycast.py: four rays from 10 cm below a level quadcopter hovering at 1 m."""
    pos = np.array([[0, 0, 1.0], [0.5, 0, 0.9]])
    quat = np.tile([0, 0, 0, 1.0], (2, 1))
    dirs = np.array([[1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, -1]], np.float32)
    out = oracle.raycast(pos, quat, 0, np.array([0, 0, -0.1], np.float32), dirs, body=True)
    assert list(out["object"]) == [1, -1, -1, 2]          # the neighbour, nothing, nothing, the ground
    np.testing.assert_allclose(out["pos"][0], [0.44, 0, 0], atol=1e-6)      # neighbour's side wall at x = 0.5 - rc
    np.testing.assert_allclose(out["pos"][3], [0, 0, -0.4], atol=1e-6)      # ground top 0.5, start 0.9
    np.testing.assert_allclose(out["dist"], [0.44, 0, 0, 0.4], atol=1e-6)
