"""The CPU oracle against the golden vectors captured from the reference's own Python
(tools/gen_golden.py).  CPU only."""
import glob
import os

import numpy as np
import pytest

import oracle

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_derived_parameters():
    # SURVEY.md 8a row P (Quadcopter.calculate_parameters, Quadcopter.py:153-168)
    d = oracle.derived()
    assert d["GravityForce"] == pytest.approx(0.26487, rel=1e-12)
    assert d["HoverRPM"] == pytest.approx(14475.809, abs=1e-3)
    assert d["MaxRPM"] == pytest.approx(21713.714, abs=1e-3)
    assert d["MaxThrust"] == pytest.approx(0.5959575, rel=1e-9)
    assert d["MaxXYTorque"] == pytest.approx(8.3649e-3, rel=1e-4)
    assert d["MaxZTorque"] == pytest.approx(7.4872e-3, rel=1e-4)
    assert d["GroundEffectHClip"] == pytest.approx(0.0377637, rel=1e-5)


def test_survey_known_answers():
    # SURVEY.md 8c anchors; inputs are python floats there, float32 here, hence the 1e-3 rpm slack
    # on the values that depend on the float32 truncation of 0.01 / 0.1
    c = oracle.Controller()
    r = c.vel_control([0, 0, 0], [0, 0, 0], [0, 0, 0], [0.5, 0, 0])
    np.testing.assert_allclose(r, [14214.56519004, 15073.76519004, 15073.76519004, 14214.56519004], atol=1e-6)
    r = c.vel_control([0.01, 0, 0], [0, 0.01, 0], [0, 0.1, 0], [0.5, 0, 0])
    np.testing.assert_allclose(r, [14110.11534986, 14969.31534986, 14969.31534986, 14110.11534986], atol=1e-3)
    c = oracle.Controller()
    r = c.pos_control([0, 0, 1], [0, 0, 0], [0, 0, 0.3], [0, 0, 0], [1, 1, 2])
    np.testing.assert_allclose(r, [14728.22912486, 16041.00651222, 16446.62912486, 16041.00651222], atol=2e-3)
    for args, want in [((0.027 * 9.81, 0, 0, 0), [14475.80915296] * 4),
                       ((0.027, 1.4e-5, 0, 0), [4664.2590609, 4664.2590609, 4578.88702636, 4578.88702636]),
                       ((0.027, 3e-3, 0, 0), [9211.17475125, 9211.17475125, 0, 0]),
                       ((0.1, 0, -5e-3, 1e-4), [13972.41181714, 0, 0, 14269.7090935]),
                       ((0, 1e-3, 1e-3, 0), [0, 6129.96615759, 0, 0])]:
        np.testing.assert_allclose(oracle.nnls_rpm(*args)[0], want, atol=1e-6)


def test_F1_quadcontrol_cascade():
    d = np.load(os.path.join(G, "F1_quadcontrol.npz"))
    for m, mode in enumerate([str(x) for x in d["modes"]]):
        for i in range(d["rpm"].shape[1]):
            c = oracle.Controller()
            for k in range(d["rpm"].shape[2]):
                a = {x: d[x][i, k] for x in ("pos", "vel", "ori", "angvel", "target")}
                if mode == "pos":
                    r = c.pos_control(a["pos"], a["vel"], a["ori"], a["angvel"], a["target"])
                elif mode == "vel":
                    r = c.vel_control(a["vel"], a["ori"], a["angvel"], a["target"])
                elif mode == "accel":
                    r = c.accel_control(a["target"].astype(np.float64), a["ori"], a["angvel"])
                else:
                    r = c.attitude_control((a["target"] * np.float32(0.3)).astype(np.float64), a["ori"], a["angvel"])
                np.testing.assert_allclose(r, d["rpm"][m, i, k], rtol=0, atol=1e-8, err_msg="%s %d %d" % (mode, i, k))


def test_F2_nnls_mixer():
    d = np.load(os.path.join(G, "F2_nnls.npz"))
    assert d["nnls_branch"].mean() > 0.3
    for w, want in zip(d["wrench"], d["rpm"]):
        np.testing.assert_allclose(oracle.nnls_rpm(*w)[0], want, rtol=0, atol=1e-8)


def test_F3_adjacency_bit_exact():
    d = np.load(os.path.join(G, "F3_adjacency.npz"))
    n = 0
    for k in d.files:
        if k.endswith("_pos"):
            key = k[:-4]
            R = float(key.split("_R")[1].replace("p", "."))
            A = oracle.adjacency(d[k], R).astype(np.uint8)
            assert np.array_equal(A, d[key + "_A"]), key
            n += 1
        elif k.startswith("pairs") and k.endswith("_a"):
            key = k[:-2]
            R = float(key.split("_R")[1].replace("p", "."))
            a, b = d[key + "_a"], d[key + "_b"]
            pos = np.stack([a, b], 1)  # (P,2,3): each pair is its own 2-agent env
            sw_A = np.zeros((len(a), 2, 2), np.float32)
            oracle.lib().orc_adjacency_batch(len(a), 2, oracle._f(np.ascontiguousarray(pos)), R, oracle._f(sw_A))
            assert np.array_equal(sw_A[:, 0, 1].astype(np.uint8), d[key + "_adj"]), key
            assert 0.2 < d[key + "_adj"].mean() < 0.8   # the sweep really straddles the threshold
            n += 1
    assert n >= 30


F6 = sorted(glob.glob(os.path.join(G, "F6_step_N*.npz")))


def _swarm(d, N, literal=True):
    """literal: the fixture's own contact settings -- since round 5 every F6 file is generated at the model's DEFINITION (rows swept to
    convergence, no closed forms; tools/gen_golden.py LITERAL) and carries them; False: the library's defaults (cap of 10, closed forms)."""
    sw = oracle.OracleSwarm(1, N)
    if literal and "solver_iters" in d.files:
        sw.p.solver_iters = int(d["solver_iters"])
        if "rest_shortcut" in d.files:
            sw.p.rest_shortcut = int(d["rest_shortcut"])
    sw.set_state(pos=d["start"].astype(np.float64), euler=d["ori0"], vel=np.zeros((N, 3)), angvel=np.zeros((N, 3)))
    return sw


def _state(sw):
    return np.concatenate([sw.pos[0], sw.quat[0], sw.vel[0], sw.angvel[0]], 1)


@pytest.mark.parametrize("path", F6, ids=[os.path.basename(p)[8:-4] for p in F6])
def test_F6_reference_step_teacher_forced(path):
    """Each reference step reproduced from the reference's own previous state: force assembly
    (controller -> rotor forces, ground effect, drag, downwash) and the observation/adjacency outputs."""
    d = np.load(path)
    name = os.path.basename(path)[8:-4]
    N, atype = int(name.split("_")[0][1:]), name.split("_", 1)[1]
    sw = _swarm(d, N)
    K, D = int(d["K_HOPS"]), int(d["D"])
    if atype == "set_control":
        # the fixture really enters the NNLS branch of nnlsRPM (Quadcopter.py:204-207): an NNLS solution has a rotor at 0
        nnls = sum(int((oracle.set_control(a) == 0).any()) for a in d["actions"].reshape(-1, 4))
        assert nnls >= 0.3 * d["actions"].shape[0] * N, nnls
    for t in range(d["actions"].shape[0]):
        sw.step(d["actions"][t], atype)
        w, wr = sw.wrench[0], d["wrench"][t]
        scale = np.abs(wr[:, :3]).max(1, keepdims=True) + 1e-3
        assert (np.abs(w[:, :3] - wr[:, :3]) / scale).max() < 2e-5, (t, "force")    # downwash is float32 in the reference
        assert np.abs(w[:, 3:] - wr[:, 3:]).max() < 1e-9, (t, "torque")
        assert np.abs(_state(sw) - d["state"][t]).max() < 2e-6, t
        s = d["state"][t]
        sw.pos[0], sw.quat[0], sw.vel[0], sw.angvel[0] = s[:, 0:3], s[:, 3:7], s[:, 7:10], s[:, 10:13]
        # newest observation slice and adjacency from the (now identical) state
        o = sw.observe()
        X = np.concatenate([o["pos"][0], o["vel"][0]], 1) if D == 6 else \
            np.concatenate([o["pos"][0], o["euler"][0], o["vel"][0], o["angvel"][0]], 1)
        np.testing.assert_allclose(X, d["X"][t][0], rtol=0, atol=1e-6)
        assert np.array_equal(sw.adjacency(float(d["COMM_RANGE"]))[0].astype(np.uint8), d["A"][t][0])


@pytest.mark.parametrize("path", F6, ids=[os.path.basename(p)[8:-4] for p in F6])
def test_F6_reference_step_free_running(path):
    d = np.load(path)
    name = os.path.basename(path)[8:-4]
    N, atype = int(name.split("_")[0][1:]), name.split("_", 1)[1]
    if name in ("N12_set_control", "N3_set_control"):
        # open-loop wrenches: the bodies tumble into the ground around step 90.  From there the free-running comparison
        # is chaotic (1e-9 before the first impact, up to 4e-2 in the angular velocities after it: the two runs differ
        # by the float32 ulp of the downwash term, and a contact solve that stops on convergence turns that into a
        # different sweep count now and then); every step of these trajectories is checked by the teacher-forced test
        pytest.skip("tumbling + ground impacts: chaotic, covered by the teacher-forced test")
    sw = _swarm(d, N)
    worst = 0.0
    for t in range(d["actions"].shape[0]):
        sw.step(d["actions"][t], atype)
        if name == "N12_set_speeds" and d["state"][t][:, 2].min() < 0.6:
            break       # round 5 (converged sweeps in the fixture): from the first ground impact on this open-loop run is chaotic too (2.9e-3 by step 200)
        worst = max(worst, np.abs(_state(sw) - d["state"][t]).max())
    assert worst < 1e-3 and t > 60, (worst, t)


def test_F6_default_settings_distance_from_the_literal_model():
    """the goldens are the model's definition (converged sweeps, no closed forms); what the library ships by default
    -- at most 10 sweeps, closed forms for flat bodies -- is held against them HERE as a stated distance, teacher-forced along every
    F6 trajectory: a step whose bodies are all clear of the ground is the same step (2e-6: no contact row is touched); a step with a
    body near the ground is within 3e-3 at the 99th percentile, 3e-4 on average, 0.1 at worst (m/s, rad/s: bodies tumbling on the
    ground under open-loop thrust whose sweeps the cap cuts short; measured 1.9e-3 / 1.5e-4 / 6.5e-2 over 1 245 such body-steps).  A change of the shipped solver
    moves these numbers, never a file under tests/golden/."""
    near, clear = [], []
    for path in F6:
        d = np.load(path)
        name = os.path.basename(path)[8:-4]
        N, atype = int(name.split("_")[0][1:]), name.split("_", 1)[1]
        assert int(d["solver_iters"]) == 50 and int(d["rest_shortcut"]) == 0, name
        sw = _swarm(d, N, literal=False)
        assert sw.p.solver_iters == oracle.default_params().solver_iters == 10 and sw.p.rest_shortcut == 1
        for t in range(d["actions"].shape[0]):
            pre_z = sw.pos[0][:, 2].copy()
            sw.step(d["actions"][t], atype)
            s = d["state"][t]
            e = np.abs(_state(sw) - s).max(1)
            low = pre_z < 0.6
            near.append(e[low]); clear.append(e[~low])
            sw.pos[0], sw.quat[0], sw.vel[0], sw.angvel[0] = s[:, 0:3], s[:, 3:7], s[:, 7:10], s[:, 10:13]
    near, clear = np.concatenate(near), np.concatenate(clear)
    assert near.size > 500 and clear.size > 10000
    assert clear.max() < 2e-6, clear.max()
    assert np.quantile(near, 0.99) < 3e-3 and near.mean() < 3e-4 and near.max() < 0.1, (np.quantile(near, 0.99), near.mean(), near.max())


def test_F6_step_none_and_touchdown():
    d = np.load(os.path.join(G, "F6_step_none_touchdown.npz"))
    sw = _swarm(d, 3)
    for t in range(d["state"].shape[0]):
        sw.step(None, None)
        assert np.abs(sw.wrench).max() == 0.0       # MRS.py:243-253: no set_actions => no rotor/aero forces
        np.testing.assert_allclose(_state(sw), d["state"][t], rtol=0, atol=1e-9)
    assert sw.pos[0, 0, 2] < 0.56 and abs(sw.vel[0, 0, 2]) < 0.05      # came to rest on the ground top (z = 0.5)


def _teacher_forced_errors(d, iters, closed_forms=None):
    N = d["start"].shape[0]
    sw = _swarm(d, N)
    sw.p.solver_iters = iters
    if closed_forms is not None:
        sw.p.rest_shortcut = int(closed_forms)
    sw.p.pair_contact = 0           # the harness integrates every body on its own (oracle.integrate)
    errs = []
    for t in range(d["actions"].shape[0]):
        sw.step(d["actions"][t], "set_speeds")
        s = d["state"][t]
        errs.append(np.abs(_state(sw) - s).max(1))
        sw.pos[0], sw.quat[0], sw.vel[0], sw.angvel[0] = s[:, 0:3], s[:, 3:7], s[:, 7:10], s[:, 10:13]
    return np.stack(errs)


def test_F6c_default_sweep_cap_against_converged_contact():
    """a fixture generated at a CONVERGED sweep count (50: bodies tumbling on the ground under rotor thrust, the
    hard case of the sequential-impulse solve), and the oracle stepped at the library's DEFAULT cap teacher-forced along it.
    Stated accuracy of the default (10 sweeps): 99 % of the body-steps within 2e-3, mean within 1.5e-4 of the converged solve
    (m/s, rad/s; measured 9.3e-4 / 6.3e-5).  The bounds sit between 10 sweeps and 8 (4.0e-3 / 2.2e-4; 6 sweeps: 2.1e-2 /
    8.2e-4; 12: 2.4e-4 / 2.1e-5; 20: 7.6e-7 / 3.7e-7), so lowering the default cap fails HERE instead of being absorbed by
    regenerated goldens (round 3 went 10 -> 6 that way)."""
    d = np.load(os.path.join(G, "F6c_tumbling_converged.npz"))
    assert int(d["solver_iters"]) == 50
    default = oracle.default_params().solver_iters
    assert default >= 10
    e = _teacher_forced_errors(d, default)
    assert (d["state"][-1][:, 2] < 0.6).mean() > 0.7                       # they did end on the ground
    assert np.quantile(e, 0.99) < 2e-3 and e.mean() < 1.5e-4, (np.quantile(e, 0.99), e.mean())
    # the fixture's own settings (50 sweeps, the rows only: no closed forms) reproduce it (float32 ulps of the downwash term)
    assert _teacher_forced_errors(d, 50, closed_forms=False).max() < 1e-6
    # with the closed forms for flat bodies (the default) the 50-sweep run stays within the model's flatness bound of it:
    # a body within a microradian of flat is not levelled (<= 1e-6 / dt = 1e-4 rad/s)
    assert _teacher_forced_errors(d, 50, closed_forms=True).max() < 1e-4
    e8 = _teacher_forced_errors(d, 8)
    assert np.quantile(e8, 0.99) > 2e-3 and e8.mean() > 1.5e-4                # the test does see a lower cap


def test_F5b_spawn_rejection_replay(golden_dir):
    """Row R: the reference's generate_start_pos (MRS.py:127-154) driven by a replay distribution -- per-agent and joint
    forms, torch.mode ties planted -- against the oracle's restatement: final layout bit for bit, rounds consumed equal."""
    d = np.load(os.path.join(golden_dir, "F5b_spawn_replay.npz"))
    names = sorted(k[:-5] for k in d.files if k.endswith("_cand"))
    assert len(names) == 48 and max(d[n + "_cand"].shape[0] for n in names) > 50
    ties = 0
    for n in names:
        cand = d[n + "_cand"]
        pos, used = oracle.spawn_from(cand)
        assert used == cand.shape[0], n
        assert np.array_equal(pos, d[n + "_final"]), n
        ties += int(n.endswith("_0") or n.endswith("_1"))
        # one round short: the layout still collides and the oracle says so
        if cand.shape[0] > 1:
            assert oracle.spawn_from(cand[:-1])[1] == -1, n
    assert ties >= 12


def test_F7_reynolds_expert(golden_dir):
    """oracle.reynolds vs the reference's Reynolds.forward_batch (examples/simulating_data/helper/Reynolds.py:80-110,
    K=1).  float32; the reference's torch reductions sum over neighbours in their own order, hence not bit-exact:
    |action| <= 1 by construction, tolerance 2e-6 absolute (measured 6.6e-7 at N=64).  Coincident agents give
    0/0 = NaN -> 0 (Reynolds.py:105) in both."""
    d = np.load(os.path.join(golden_dir, "F7_reynolds.npz"))
    n = 0
    for k in d.files:
        if not k.endswith("_Xs"):
            continue
        Xs, want = d[k], d[k[:-3] + "_actions"]
        got = oracle.reynolds(Xs[..., 1])
        assert got.shape == want.shape and not np.isnan(got).any()
        assert np.abs(got - want).max() < 2e-6, k
        assert np.array_equal(got == 0, want == 0), k          # the NaN -> 0 rows coincide
        n += 1
    assert n == 6      # four K = 1 cases + the K = 2 and K = 3 cases (the controller reads hops 0 and 1 only, Reynolds_Node.py:30)
