#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE's own Python.

Runs only in the build container (needs /root/reference); the fixtures it writes are plain
.npz data (inputs + expected outputs) and are committed.  Nothing from the reference's source
is copied: the reference package is imported from where it lies.

The reference cannot be imported as-is here: `pybullet`, `gym` and `ray` are not installed and
cannot be (no network).  Two harnesses are used:

  * inert stand-in modules for gym / ray / pybullet so that `import mrsgym` succeeds; every
    function that never touches `p.*` then runs unmodified (QuadControl cascade, nnlsRPM,
    MRS.calc_A, history deques, generate_start_pos).  -> F1..F5

  * "fake bullet": the stand-in `pybullet` module is backed by the build's OWN CPU oracle
    integrator (oracle.integrate), so the reference's real MRS.step()/reset() Python --
    controller, force assembly, call ordering, dynamics(), history, adjacency, callbacks --
    runs end to end.  This pins everything EXCEPT the Bullet integrator/contact itself, which
    stays "parity unpinned" (pybullet absent).  -> F6

Usage:  python tools/gen_golden.py [--out tests/golden]
"""
import argparse
import os
import sys
import types
from collections import deque

import numpy as np

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = os.environ.get("MRS_REFERENCE", "/root/reference")

import oracle  # noqa: E402  (the fake-bullet integrator)


# --------------------------------------------------------------------------- stand-ins
def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


# The MODEL'S DEFINITION for every F6 fixture (round 5): the contact rows swept to convergence (50 sweeps: they stop on
# their own tolerance long before) and no closed forms for flat bodies -- the literal rows.  What the library ships by default (a cap
# of 10 sweeps, closed forms) is an approximation of that, held against these fixtures as a stated DISTANCE by the tests
# (tests/test_oracle_golden.py, tests/test_gpu_parity.py); a change of the shipped solver must never touch a file under tests/golden/.
LITERAL = dict(solver_iters=50, rest_shortcut=0)


class FakeBullet:
    """Just enough of the pybullet C-API surface the reference's hot path calls
    (SURVEY.md 8b "Bottom"), backed by oracle.integrate."""
    DIRECT, GUI, LINK_FRAME, WORLD_FRAME, COV_ENABLE_GUI = 2, 1, 1, 2, 1

    def __init__(self):
        self.params = oracle.default_params()
        for k, v in LITERAL.items():
            setattr(self.params, k, v)
        self.bodies = {}
        self.next_id = 0
        self.debug_id = 0

    # -- setup
    def connect(self, mode, **k):
        return 0

    def configureDebugVisualizer(self, **k):
        pass

    def setGravity(self, gravX, gravY, gravZ, physicsClientId=0):
        self.params.gravity = -gravZ

    def setTimeStep(self, timeStep, physicsClientId=0):
        self.params.dt = timeStep

    def setRealTimeSimulation(self, *a, **k):
        pass

    def resetSimulation(self, *a, **k):
        self.bodies.clear()

    def disconnect(self, *a, **k):
        pass

    def removeBody(self, *a, **k):
        pass

    def loadURDF(self, fileName, basePosition, baseOrientation, physicsClientId=0):
        uid = self.next_id
        self.next_id += 1
        quad = os.path.basename(fileName).startswith("cf2")
        self.bodies[uid] = dict(quad=quad, pos=np.array(basePosition, float), quat=np.array(baseOrientation, float),
                                vel=np.zeros(3), angvel=np.zeros(3), fb=np.zeros(3), tb=np.zeros(3))
        return uid

    # -- state
    def resetBasePositionAndOrientation(self, uid, posObj, ornObj, physicsClientId=0):
        b = self.bodies[uid]
        b["pos"] = np.array(posObj, float)
        b["quat"] = np.array(ornObj, float)

    def resetBaseVelocity(self, uid, linearVelocity, angularVelocity, physicsClientId=0):
        b = self.bodies[uid]
        b["vel"] = np.array(linearVelocity, float)
        b["angvel"] = np.array(angularVelocity, float)

    def getBasePositionAndOrientation(self, uid, physicsClientId=0):
        b = self.bodies[uid]
        return tuple(b["pos"].tolist()), tuple(b["quat"].tolist())

    def getBaseVelocity(self, uid, physicsClientId=0):
        b = self.bodies[uid]
        return tuple(b["vel"].tolist()), tuple(b["angvel"].tolist())

    def _link_offset(self, link):
        p = self.params
        if link < 4:
            return np.array([p.prop_x[link], p.prop_y[link], p.prop_z[link]])
        return np.zeros(3)

    def getLinkStates(self, uid, linkIndices, computeLinkVelocity=0, computeForwardKinematics=0, physicsClientId=0):
        # every entry padded to length 4 so the reference's np.array(...) of the (ragged) real
        # return value also works on numpy >= 1.24; only [link, 0][2] (world z of the link COM) is read
        b = self.bodies[uid]
        R = oracle.quat_to_matrix(b["quat"] / np.linalg.norm(b["quat"]))
        out = []
        for li in linkIndices:
            w = b["pos"] + R @ self._link_offset(li)
            pos4 = (w[0], w[1], w[2], 0.0)
            out.append(tuple([pos4] + [(0.0, 0.0, 0.0, 1.0)] * 7))
        return tuple(out)

    # -- forces (LINK_FRAME at the link COM, posObj = 0)
    def applyExternalForce(self, uid, linkIndex, forceObj, posObj, flags, physicsClientId=0):
        assert flags == self.LINK_FRAME
        b = self.bodies[uid]
        f = np.array([float(x) for x in forceObj])
        b["fb"] += f
        b["tb"] += np.cross(self._link_offset(linkIndex), f)

    def applyExternalTorque(self, uid, linkIndex, torqueObj, flags, physicsClientId=0):
        assert flags == self.LINK_FRAME
        self.bodies[uid]["tb"] += np.array([float(x) for x in torqueObj])

    def stepSimulation(self, physicsClientId=0):
        for b in self.bodies.values():
            b["last_wrench"] = np.concatenate([b["fb"], b["tb"]])
            if b["quad"]:
                oracle.integrate(self.params, b["pos"], b["quat"], b["vel"], b["angvel"], b["fb"], b["tb"])
            b["fb"][:] = 0
            b["tb"][:] = 0

    # -- debug drawing (MRS.step -> Environment.draw_links)
    def addUserDebugLine(self, **k):
        self.debug_id += 1
        return self.debug_id

    def removeUserDebugItem(self, **k):
        pass


def import_reference(fake):
    for k in [k for k in sys.modules if k == "mrsgym" or k.startswith("mrsgym.")]:
        del sys.modules[k]
    pb = _stub("pybullet")
    for name in dir(fake):
        if not name.startswith("_"):
            setattr(pb, name, getattr(fake, name))
    gym = _stub("gym", Env=type("Env", (), {"__init__": lambda self, *a, **k: None}))
    gym.spaces = _stub("gym.spaces", Box=lambda low, high, dtype=None: types.SimpleNamespace(low=low, high=high))
    gym.envs = _stub("gym.envs")
    gym.envs.registration = _stub("gym.envs.registration", register=lambda **k: None)
    _stub("ray"); _stub("ray.rllib"); _stub("ray.rllib.env")
    _stub("ray.rllib.env.multi_agent_env", MultiAgentEnv=type("MultiAgentEnv", (), {}))
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import mrsgym
    return mrsgym


# --------------------------------------------------------------------------- fixtures
def gen_F1(mrsgym, out):
    """QuadControl cascade: 256 random tuples x {pos,vel,accel,ori} x 5 consecutive calls."""
    import torch
    from mrsgym.QuadControl import QuadControl
    rng = np.random.default_rng(101)
    n, calls = 256, 5
    attrs = {"Mass": 0.027, "Kf": 3.16e-10}
    modes = ["pos", "vel", "accel", "ori"]
    inp = {k: np.zeros((n, calls, 3), np.float32) for k in ("pos", "vel", "ori", "angvel", "target")}
    rpm = np.zeros((len(modes), n, calls, 4))
    for i in range(n):
        for c in range(calls):
            inp["pos"][i, c] = rng.normal(0, 2, 3)
            inp["vel"][i, c] = rng.normal(0, 1.5, 3)
            scale = 0.3 if i % 4 else 1.2     # a quarter of the cases at large attitude
            inp["ori"][i, c] = rng.uniform(-1, 1, 3) * np.array([scale, scale, np.pi])
            inp["angvel"][i, c] = rng.normal(0, 1.0, 3)
            inp["target"][i, c] = rng.normal(0, 1.5, 3)
    for m, mode in enumerate(modes):
        for i in range(n):
            ctl = QuadControl(dict(attrs))
            for c in range(calls):
                t = {k: torch.tensor(inp[k][i, c]) for k in inp}     # float32 tensors, as Object.get_* returns
                if mode == "pos":
                    r = ctl.pos_control(pos=t["pos"], vel=t["vel"], ori=t["ori"], angvel=t["angvel"], target_pos=t["target"])
                elif mode == "vel":
                    r = ctl.vel_control(vel=t["vel"], ori=t["ori"], angvel=t["angvel"], target_vel=t["target"])
                elif mode == "accel":
                    r = ctl.accel_control(ori=t["ori"], angvel=t["angvel"], target_accel=t["target"])
                else:
                    r = ctl.attitude_control(target_accel=np.array([0., 0., 9.81]), target_ori=t["target"] * 0.3,
                                             ori=t["ori"], angvel=t["angvel"])
                rpm[m, i, c] = r
    np.savez_compressed(os.path.join(out, "F1_quadcontrol.npz"), modes=np.array(modes), rpm=rpm, **inp)
    # the survey's known-answer anchors (python-float inputs)
    ctl = QuadControl(dict(attrs))
    a1 = ctl.vel_control(vel=np.zeros(3), ori=np.zeros(3), angvel=np.zeros(3), target_vel=[0.5, 0, 0])
    return a1


def gen_F2(mrsgym, out):
    """nnlsRPM: 512 random wrenches, >= half with a negative direct-inverse component."""
    from mrsgym.Quadcopter import nnlsRPM
    rng = np.random.default_rng(202)
    c = 1 / np.sqrt(2)
    A = np.array([[1, 1, 1, 1], [c, c, -c, -c], [-c, c, c, -c], [-1, 1, -1, 1]])
    Ainv = np.linalg.inv(A)
    kf, km, L = 3.16e-10, 7.94e-12, 0.0397
    bc = np.array([1 / kf, 1 / (kf * L), 1 / (kf * L), 1 / km])
    w = np.zeros((512, 4)); rpm = np.zeros((512, 4)); neg = np.zeros(512, bool)
    i = 0
    while i < 512:
        big = i % 2 == 1
        thrust = rng.uniform(0, 0.6)
        tq = rng.normal(0, 1, 3) * (np.array([8e-3, 8e-3, 7e-3]) if big else np.array([4e-4, 4e-4, 2e-4]))
        try:
            r = nnlsRPM(thrust, tq[0], tq[1], tq[2], 0, 0, 0, A, Ainv, bc)
        except RuntimeError:        # scipy raises when maxiter is hit; the reference would crash
            continue
        w[i] = [thrust, *tq]; rpm[i] = r
        neg[i] = (Ainv @ (w[i] * bc)).min() < 0
        i += 1
    np.savez_compressed(os.path.join(out, "F2_nnls.npz"), wrench=w, rpm=rpm, nnls_branch=neg)
    return neg.mean()


def _bare_mrs(mrsgym, N, K, comm_range):
    m = mrsgym.MRS.__new__(mrsgym.MRS)      # skip __init__ (needs a sim); methods under test only read these
    m.N_AGENTS, m.K_HOPS, m.COMM_RANGE = N, K, comm_range
    m.X, m.A = deque([]), deque([])
    return m


def gen_F3(mrsgym, out):
    """MRS.calc_A for N in {3,64,256}, COMM_RANGE in {2.5, 5, 0.7, inf}, incl. near-threshold pairs."""
    import torch
    rng = np.random.default_rng(303)
    data = {}
    for N in (3, 64, 256):
        for cr in (2.5, 5.0, 0.7, float("inf")):
            side = int(np.ceil(np.sqrt(N)))
            g = np.stack(np.meshgrid(np.arange(side), np.arange(side)), -1).reshape(-1, 2)[:N] - side / 2
            pos = np.concatenate([g + rng.uniform(-.2, .2, (N, 2)), rng.uniform(1, 3, (N, 1))], 1).astype(np.float32)
            if np.isfinite(cr) and N >= 3:
                # plant pairs exactly at / one ulp around the threshold
                pos[1] = pos[0] + np.array([cr, 0, 0], np.float32)
                d = np.float32(cr) / np.sqrt(np.float32(3))
                pos[2] = pos[0] - np.array([d, d, d], np.float32)
            m = _bare_mrs(mrsgym, N, 0, cr)
            m.env = types.SimpleNamespace(get_pos=lambda pos=pos: torch.tensor(pos))
            A = m.calc_A().numpy()
            key = "N%d_R%s" % (N, str(cr).replace(".", "p"))
            data[key + "_pos"] = pos
            data[key + "_A"] = A.astype(np.uint8)
    # dense near-threshold sweep: many random pairs within a few ulp of the range
    for cr in (2.5, 5.0, 0.7):
        P = 4096
        a = rng.uniform(-3, 3, (P, 3)).astype(np.float32)
        u = rng.normal(size=(P, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
        b = (a.astype(np.float64) + u * cr * (1 + rng.integers(-3, 4, (P, 1)) * 6e-8)).astype(np.float32)
        res = np.zeros(P, np.uint8)
        for k in range(P):
            m = _bare_mrs(mrsgym, 2, 0, cr)
            pk = np.stack([a[k], b[k]])
            m.env = types.SimpleNamespace(get_pos=lambda pk=pk: torch.tensor(pk))
            res[k] = m.calc_A().numpy()[0, 1]
        key = "pairs_R%s" % str(cr).replace(".", "p")
        data[key + "_a"], data[key + "_b"], data[key + "_adj"] = a, b, res
    # the same near-threshold planting inside full N=64 / N=256 swarms (the (N,N,3) expand path)
    for N in (64, 256):
        for cr in (2.5, 5.0):
            for rep in range(4):
                h = N // 2
                a = rng.uniform(-3, 3, (h, 3)).astype(np.float32)
                u = rng.normal(size=(h, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
                b = (a.astype(np.float64) + u * cr * (1 + rng.integers(-3, 4, (h, 1)) * 6e-8)).astype(np.float32)
                pos = np.concatenate([a, b]).astype(np.float32)
                m = _bare_mrs(mrsgym, N, 0, cr)
                m.env = types.SimpleNamespace(get_pos=lambda pos=pos: torch.tensor(pos))
                key = "planted%d_N%d_R%s" % (rep, N, str(cr).replace(".", "p"))
                data[key + "_pos"] = pos
                data[key + "_A"] = m.calc_A().numpy().astype(np.uint8)
    np.savez_compressed(os.path.join(out, "F3_adjacency.npz"), **data)


def gen_F4(mrsgym, out):
    """History semantics of calc_Xk / calc_Ak for K in {0,1,3}: newest-first, X padded with copies,
    A padded with zeros (MRS.py:87-114)."""
    import torch
    rng = np.random.default_rng(404)
    data = {}
    N, D, T = 5, 6, 7
    for K in (0, 1, 3):
        seqX = rng.normal(size=(T, N, D)).astype(np.float32)
        seqP = rng.uniform(-2, 2, (T, N, 3)).astype(np.float32)
        m = _bare_mrs(mrsgym, N, K, 2.0)
        it = {"t": 0}
        m.state_fn = None
        m.env = types.SimpleNamespace(get_X=lambda fn: torch.tensor(seqX[it["t"]]),
                                      get_pos=lambda: torch.tensor(seqP[it["t"]]))
        Xk, Ak = [], []
        for t in range(T):
            it["t"] = t
            Xk.append(m.calc_Xk().numpy().copy())
            if t >= 1:                      # reset() computes X only; A first appears in step()
                Ak.append(m.calc_Ak().numpy().copy())
        data["K%d_seqX" % K], data["K%d_seqP" % K] = seqX, seqP
        data["K%d_Xk" % K], data["K%d_Ak" % K] = np.stack(Xk), np.stack(Ak)
    np.savez_compressed(os.path.join(out, "F4_history.npz"), **data)


def gen_F5(mrsgym, out):
    """generate_start_pos with the default spawn distribution (property fixture; RNG not bit-matched)."""
    import torch
    data = {}
    for N in (3, 12, 32):
        outs = []
        for seed in range(4):
            torch.manual_seed(seed)
            m = _bare_mrs(mrsgym, N, 0, float("inf"))
            m.AGENT_RADIUS = 0.3
            m.START_POS = m.default_spawn_dist()
            outs.append(m.generate_start_pos().numpy())
        data["N%d" % N] = np.stack(outs)
    torch.manual_seed(0)
    m = _bare_mrs(mrsgym, 16, 0, float("inf"))
    m.START_ORI = torch.tensor([0, 0, -np.pi / 2, 0, 0, np.pi / 2]).expand(16, -1)
    data["ori_N16"] = torch.stack([m.generate_start_ori() for _ in range(8)]).numpy()
    np.savez_compressed(os.path.join(out, "F5_spawn.npz"), **data)


class _Replay:
    """A stand-in START_POS "distribution" (README.md:71-74) that hands back pre-drawn samples in order, so that the
    reference's rejection loop becomes a deterministic function of a sample stream."""

    def __init__(self, samples, events):
        self.samples, self.k, self.events = samples, 0, events

    def sample(self):
        s = self.samples[self.k]
        self.events.append(("sample", self.k))
        self.k += 1
        return s.clone()            # generate_start_pos writes into what it gets (MRS.py:148,151)


def gen_F5b(mrsgym, out):
    """Row R, deterministic: MRS.generate_start_pos (MRS.py:127-154) driven by a replay distribution, per-agent
    ((3,) samples, :132-134, :146-148) and joint ((N,3) samples, :149-151) forms, incl. torch.mode ties (:141).

    What is stored per case is the candidate tensor in the layout mrs_spawn_from takes -- cand[r][i] = the sample agent i
    receives if it is re-sampled in round r (round 0 = the first layout) -- rebuilt from the order in which the reference
    consumed the stream (its torch.mode picks and sample() calls are logged by pass-through wrappers; nothing of the
    reference is modified); entries the reference never used are far-away sentinels, so an implementation that flags a
    different agent ends somewhere else.  Expected: the final layout, bit for bit, and the rounds it took."""
    import torch
    rng = np.random.default_rng(505)
    data = {}
    real_mode = torch.mode
    boxes = {3: (0.45, 0.45, (1.0, 1.5)), 12: (1.0, 1.0, (1.0, 2.2)), 32: (1.5, 1.5, (1.0, 3.0))}
    names = []
    try:
        for N in (3, 12, 32):
            bx, by, bz = boxes[N]
            for form in ("agent", "joint"):
                for case in range(8):
                    def draw(shape):
                        p = np.stack([rng.uniform(-bx, bx, shape), rng.uniform(-by, by, shape), rng.uniform(bz[0], bz[1], shape)], -1)
                        return p.astype(np.float32)
                    n_stream = 4000 if form == "agent" else 400
                    stream = draw((n_stream,)) if form == "agent" else draw((n_stream, N))
                    if case == 0:
                        # planted ties: every agent collides with every other one (all counts equal) in the first layout
                        if form == "agent":
                            stream[1:N + 1] = stream[1] + rng.uniform(-0.05, 0.05, (N, 3)).astype(np.float32)
                        else:
                            stream[0] = stream[0, 0] + rng.uniform(-0.05, 0.05, (N, 3)).astype(np.float32)
                    if case == 1 and N >= 4:
                        # planted ties: disjoint colliding pairs (counts 1,1,1,1,...), the rest far apart
                        base = np.stack([np.arange(N) * 2.0, np.zeros(N), np.full(N, 1.5)], -1).astype(np.float32)
                        base[1] = base[0] + np.float32([0.3, 0, 0]); base[3] = base[2] + np.float32([0, 0.3, 0])
                        if form == "agent":
                            stream[1:N + 1] = base
                        else:
                            stream[0] = base
                    events = []
                    torch.mode = lambda *a, **k: (lambda r: (events.append(("mode", int(r[0]))), r)[1])(real_mode(*a, **k))
                    m = _bare_mrs(mrsgym, N, 0, float("inf"))
                    m.AGENT_RADIUS = 0.3
                    m.START_POS = _Replay([torch.from_numpy(x.copy()) for x in stream], events)
                    final = m.generate_start_pos().numpy().copy()
                    torch.mode = real_mode
                    # rebuild (rounds, N, 3) candidates from the event log
                    sentinel = lambda r: np.stack([1e4 + 10.0 * np.arange(N), np.full(N, 1e4 + 10.0 * r), np.zeros(N)], -1).astype(np.float32)
                    rounds = []
                    if form == "agent":
                        assert [e for e in events[:N + 1]] == [("sample", k) for k in range(N + 1)]
                        rounds.append(stream[1:N + 1].copy())
                        ev = events[N + 1:]
                    else:
                        assert events[0] == ("sample", 0)
                        rounds.append(stream[0].copy())
                        ev = events[1:]
                    flagged = []
                    i = 0
                    while i < len(ev):
                        picks = []
                        while i < len(ev) and ev[i][0] == "mode":
                            picks.append(ev[i][1]); i += 1
                        c = sentinel(len(rounds))
                        if form == "agent":
                            for idx in picks:
                                assert ev[i][0] == "sample"
                                c[idx] = stream[ev[i][1]]; i += 1
                        else:
                            assert ev[i][0] == "sample"
                            c[picks] = stream[ev[i][1]][picks]; i += 1
                        rounds.append(c)
                        f = np.full(N, -1, np.int32); f[:len(picks)] = picks
                        flagged.append(f)
                    key = "N%d_%s_%d" % (N, form, case)
                    names.append(key)
                    data[key + "_cand"] = np.stack(rounds)
                    data[key + "_final"] = final
                    data[key + "_flagged"] = np.stack(flagged) if flagged else np.zeros((0, N), np.int32)
                    d = np.linalg.norm(final[:, None] - final[None], axis=-1) + np.eye(N) * 9
                    assert d.min() >= 0.6 - 1e-6
    finally:
        torch.mode = real_mode
    np.savez_compressed(os.path.join(out, "F5b_spawn_replay.npz"), **data)
    return {k: int(data[k + "_cand"].shape[0]) for k in names}


def gen_F6(out):
    """Full MRS.step() trajectories through the fake-bullet harness."""
    import torch
    fake = FakeBullet()
    mrsgym = import_reference(fake)
    import gym

    def state_fn(quad):
        return torch.cat([quad.get_pos(), quad.get_vel()])

    def state_fn_full(quad):
        return torch.cat([quad.get_pos(), quad.get_ori(), quad.get_vel(), quad.get_angvel()])

    rng = np.random.default_rng(606)
    hover = 14475.809152959684
    cases = []
    for N, T in ((3, 200), (12, 200)):
        side = int(np.ceil(np.sqrt(N)))
        grid = np.stack(np.meshgrid(np.arange(side), np.arange(side)), -1).reshape(-1, 2)[:N] - (side - 1) / 2
        for atype in ("set_target_vel", "set_target_pos", "set_target_accel", "set_target_ori", "set_control", "set_speeds"):
            # grid at 2*AGENT_RADIUS pitch, z spread 1..3; agent 1 hovers ~1 m above agent 0 (and 5 above 4)
            # so that the downwash term is a sizeable fraction of the weight
            start = np.concatenate([grid * 0.6 + rng.uniform(-.05, .05, (N, 2)), rng.uniform(1.0, 3.0, (N, 1))], 1)
            for lo, hi in ((0, 1), (4, 5)):
                if hi < N:
                    start[lo, 2] = rng.uniform(1.0, 1.5)
                    start[hi] = start[lo] + np.array([rng.uniform(-.05, .05), rng.uniform(-.05, .05), rng.uniform(0.9, 1.2)])
            if atype == "set_control":
                start[:, 2] += 2.5                  # open loop with free-fall blocks: keep it off the ground for a while
            start = start.astype(np.float32)
            ori0 = np.concatenate([rng.uniform(-.1, .1, (N, 2)), rng.uniform(-np.pi / 2, np.pi / 2, (N, 1))], 1).astype(np.float32)
            fake.__init__()
            env = mrsgym.MRS(state_fn=state_fn_full if N == 3 else state_fn, N_AGENTS=N, K_HOPS=2, COMM_RANGE=1.0,
                             ACTION_TYPE=atype, HEADLESS=True, START_POS=torch.tensor(start))
            X0 = env.reset(ori=torch.tensor(ori0)).numpy().copy()
            acts, Xs, As, states, rewards, dones, wrenches = [], [], [], [], [], [], []
            act = None
            for t in range(T):
                if atype == "set_target_vel":
                    if t % 50 == 0:
                        act = rng.uniform(-1, 1, (N, 3))
                elif atype == "set_target_pos":
                    if t % 100 == 0:
                        act = start + rng.uniform(-1, 1, (N, 3))
                elif atype == "set_target_accel":
                    if t % 25 == 0:
                        act = rng.uniform(-2, 2, (N, 3))
                elif atype == "set_target_ori":
                    if t % 40 == 0:
                        act = rng.uniform(-.3, .3, (N, 3))
                elif atype == "set_control":
                    if t % 20 == 0:
                        act = np.concatenate([9.81 + rng.uniform(-1, 1, (N, 1)), rng.uniform(-1, 1, (N, 3))], 1)
                        # Quadcopter.py:204: scipy's nnls runs iff min(Ainv B) < 0, i.e. iff the torque terms
                        # 0.354 (|B1| + |B2|) + 0.25 |B3| (~7e5 per unit of |control|) exceed 0.25 B0 = 2.1e7 * a_thrust.
                        # Every other block: thrust 0.2..0.6 m/s^2 and torques x20 => most quadcopters take the NNLS
                        # branch (asserted in tests: nnls_branch); the block after it thrusts 2 g minus that to recover.
                        if t % 40 == 0:
                            act[:, 0] = 0.4 + 0.2 * (act[:, 0] - 9.81)
                            act[:, 1:] *= 20.0
                        else:
                            act[:, 0] += 9.81 - 0.4
                else:
                    act = hover * (1 + 0.05 * rng.uniform(-1, 1, (N, 4)))
                a32 = torch.tensor(act, dtype=torch.float32)
                try:
                    X, r, d, info = env.step(a32)
                except RuntimeError as e:       # scipy nnls maxiter => the reference itself crashes; stop the case
                    print("   reference raised at t=%d: %s" % (t, e))
                    break
                acts.append(a32.numpy().copy()); Xs.append(X.numpy().copy()); As.append(info["A"].numpy().copy())
                rewards.append(float(r)); dones.append(bool(d))
                st = []
                for ag in env.env.agents:
                    b = fake.bodies[ag.uid]
                    st.append(np.concatenate([b["pos"], b["quat"], b["vel"], b["angvel"]]))
                states.append(np.stack(st))
                wrenches.append(np.stack([fake.bodies[ag.uid]["last_wrench"] for ag in env.env.agents]))
            cases.append((N, atype))
            key = "N%d_%s" % (N, atype)
            np.savez_compressed(os.path.join(out, "F6_step_%s.npz" % key), start=start, ori0=ori0, X0=X0,
                                actions=np.stack(acts), X=np.stack(Xs), A=np.stack(As).astype(np.uint8),
                                state=np.stack(states), wrench=np.stack(wrenches), reward=np.array(rewards), done=np.array(dones),
                                K_HOPS=2, COMM_RANGE=1.0, D=X0.shape[-1], solver_iters=LITERAL["solver_iters"], rest_shortcut=LITERAL["rest_shortcut"])
            print("  F6", key, "steps", len(acts), "final z", np.stack(states)[-1][:, 2].round(3)[:4])
            env.close()
    # step(None) quirk (MRS.py:243-253) and touchdown on the ground plane
    fake.__init__()
    N = 3
    start = np.array([[0, 0, 0.56], [1, 0, 0.7], [0, 1, 1.0]], np.float32)
    env = mrsgym.MRS(state_fn=state_fn, N_AGENTS=N, K_HOPS=0, ACTION_TYPE="set_speeds", HEADLESS=True,
                     START_POS=torch.tensor(start))
    env.reset(ori=torch.tensor(np.array([[0.2, -0.1, 0.3], [0, 0, 0], [0.5, 0.4, -1.0]], np.float32)))
    states = []
    for t in range(150):
        env.step(None)
        states.append(np.stack([np.concatenate([fake.bodies[a.uid][k] for k in ("pos", "quat", "vel", "angvel")]) for a in env.env.agents]))
    np.savez_compressed(os.path.join(out, "F6_step_none_touchdown.npz"), start=start,
                        ori0=np.array([[0.2, -0.1, 0.3], [0, 0, 0], [0.5, 0.4, -1.0]], np.float32), state=np.stack(states),
                        solver_iters=LITERAL["solver_iters"], rest_shortcut=LITERAL["rest_shortcut"])
    env.close()
    del env
    import gc
    gc.collect()        # MRS.__del__ -> close() -> p.resetSimulation: let the old env go BEFORE the next one loads its bodies
    # Contact sweeps at a CONVERGED count: twelve quadcopters dropped tilted from just above the ground with rotor
    # speeds of hover +-5 % -- they tumble on the ground under thrust, the hard case of the sequential-impulse solve -- through
    # the same harness with solver_iters = 50 (pybullet's own default, BulletSim.py:34 sets none; the sweeps stop on
    # convergence well before).  tests/test_oracle_golden.py steps the oracle at its DEFAULT cap teacher-forced along this
    # trajectory: a lower default shows up there as a failing test, not as a regenerated fixture.
    fake.__init__()
    fake.params.solver_iters = 50
    fake.params.rest_shortcut = 0       # the rows themselves, swept to convergence: no closed forms in the reference trajectory
    N, T = 12, 300
    r2 = np.random.default_rng(66)
    side = 4
    grid = np.stack(np.meshgrid(np.arange(side), np.arange(side)), -1).reshape(-1, 2)[:N] - (side - 1) / 2
    start = np.concatenate([grid + r2.uniform(-.2, .2, (N, 2)), r2.uniform(0.56, 0.9, (N, 1))], 1).astype(np.float32)
    ori0 = np.concatenate([r2.uniform(-.5, .5, (N, 2)), r2.uniform(-np.pi / 2, np.pi / 2, (N, 1))], 1).astype(np.float32)
    env = mrsgym.MRS(state_fn=state_fn, N_AGENTS=N, K_HOPS=0, ACTION_TYPE="set_speeds", HEADLESS=True, START_POS=torch.tensor(start))
    env.reset(ori=torch.tensor(ori0))
    acts, states = [], []
    for t in range(T):
        a32 = torch.tensor(hover * (1 + 0.05 * r2.uniform(-1, 1, (N, 4))), dtype=torch.float32)
        env.step(a32)
        acts.append(a32.numpy().copy())
        states.append(np.stack([np.concatenate([fake.bodies[a.uid][k] for k in ("pos", "quat", "vel", "angvel")]) for a in env.env.agents]))
    np.savez_compressed(os.path.join(out, "F6c_tumbling_converged.npz"), start=start, ori0=ori0, actions=np.stack(acts),
                        state=np.stack(states), solver_iters=50)
    print("  F6c tumbling, converged sweeps: final z", np.stack(states)[-1][:, 2].round(3)[:6])
    env.close()
    fake.__init__()
    return cases


def gen_F7(out):
    """The Reynolds flocking expert the reference's data generator drives the env with
    (examples/simulating_data/helper/Reynolds.py:80-110 forward_batch, Reynolds_Node.py:26-38), as called by
    gen_data.py:33 (D=6, K=1).  Inputs in ITS layout (batch, N, D, K+1); plain torch, imported from where it lies."""
    import torch
    sys.path.insert(0, os.path.join(REF, "examples", "simulating_data"))
    from helper.Reynolds import Reynolds
    rng = np.random.default_rng(77)
    data = {}
    for N, B, D in ((3, 6, 6), (12, 5, 6), (64, 4, 6), (12, 3, 9)):
        # swarm-like inputs: positions spread over a few metres, velocities ~ 1 m/s; one env with two coincident agents
        Xs = np.concatenate([rng.normal(0, 1.5, (B, N, 3, 2)), rng.normal(0, 0.7, (B, N, D - 3, 2))], 2).astype(np.float32)
        Xs[0, 1, :3, 1] = Xs[0, 0, :3, 1]
        model = Reynolds(N=N, D=D, K=1, OUT_DIM=3)
        with torch.no_grad():
            act = model.forward(None, torch.from_numpy(Xs.copy()))
        key = "N%d_D%d" % (N, D)
        data[key + "_Xs"] = Xs
        data[key + "_actions"] = act.numpy().astype(np.float32)
    # K > 1 (Reynolds.py:89-97 aggregate hops 2..K, but Reynolds_Node.py:30 reads hops 0 and 1 only)
    for N, B, D, K in ((12, 4, 6, 2), (12, 3, 9, 3)):
        Xs = np.concatenate([rng.normal(0, 1.5, (B, N, 3, K + 1)), rng.normal(0, 0.7, (B, N, D - 3, K + 1))], 2).astype(np.float32)
        model = Reynolds(N=N, D=D, K=K, OUT_DIM=3)
        with torch.no_grad():
            act = model.forward(None, torch.from_numpy(Xs.copy()))
        key = "N%d_D%d_K%d" % (N, D, K)
        data[key + "_Xs"] = Xs
        data[key + "_actions"] = act.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(out, "F7_reynolds.npz"), **data)
    return sorted(k for k in data if k.endswith("_Xs"))


def gen_F8(out):
    """Flocking metrics of the reference's analytics (examples/simulating_data/helper/MRSAnalytics.py:13-101) on
    synthetic episode tensors X (episodes, length, N, 6), incl. an episode with two coincident agents (separation
    masks zero distances with inf, :70) -- the class as it is, fed through the `data.get_episodes()` it expects."""
    import torch
    sys.path.insert(0, os.path.join(REF, "examples", "simulating_data"))
    _stub("gym")
    from helper.MRSAnalytics import MRSAnalytics
    rng = np.random.default_rng(88)
    data = {}
    for N, EP, L in ((3, 2, 5), (12, 3, 7), (64, 2, 4)):
        X = np.concatenate([rng.normal(0, 1.5, (EP, L, N, 3)), rng.normal(0, 0.7, (EP, L, N, 3))], -1).astype(np.float32)
        X[0, 0, 1, :3] = X[0, 0, 0, :3]
        an = MRSAnalytics(types.SimpleNamespace(get_episodes=lambda X=X: {"X": torch.from_numpy(X.copy())}))
        key = "N%d" % N
        data[key + "_X"] = X
        for name in ("separation", "cohesion", "dist_to_leader", "vel_stddev", "vel_mag", "vel_leader_alignment"):
            data[key + "_" + name] = getattr(an, name)().numpy()
        data[key + "_cohesion_noleader"] = an.cohesion(exclude_leader=True).numpy()
        for name in ("separation_avg", "cohesion_avg", "vel_stddev_avg", "vel_mag_avg", "vel_leader_alignment_avg"):
            data[key + "_" + name] = np.float32(getattr(an, name)())
    np.savez_compressed(os.path.join(out, "F8_flock_metrics.npz"), **data)
    return sorted(k for k in data if k.endswith("_X"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    ap.add_argument("--only", default=None, help="generate one fixture family only, e.g. F7")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    import scipy
    print("numpy", np.__version__, "scipy", scipy.__version__)
    if args.only == "F7":
        print("F7", gen_F7(args.out))
        return
    if args.only == "F8":
        import_reference(FakeBullet())
        print("F8", gen_F8(args.out))
        return
    if args.only == "F6":
        print("F6", gen_F6(args.out))
        return
    if args.only == "F5b":
        print("F5b rounds", gen_F5b(import_reference(FakeBullet()), args.out))
        return
    mrsgym = import_reference(FakeBullet())
    print("F1 anchor", gen_F1(mrsgym, args.out))
    print("F2 nnls-branch fraction", gen_F2(mrsgym, args.out))
    gen_F3(mrsgym, args.out); print("F3 done")
    gen_F4(mrsgym, args.out); print("F4 done")
    gen_F5(mrsgym, args.out); print("F5 done")
    print("F5b rounds", gen_F5b(mrsgym, args.out))
    print("F6", gen_F6(args.out))
    print("F7", gen_F7(args.out))
    print("F8", gen_F8(args.out))
    with open(os.path.join(args.out, "README.md"), "w") as f:
        f.write("Golden fixtures generated by tools/gen_golden.py from the reference's own Python\n"
                "(numpy %s, scipy %s).  Data only: inputs and expected outputs.\n"
                "F1 QuadControl cascade, F2 nnlsRPM, F3 MRS.calc_A, F4 history deques, F5 spawn (properties),\n"
                "F5b MRS.generate_start_pos driven by a replay distribution (candidates, final layout, picks per round),\n"
                "F6 full MRS.step() trajectories with pybullet replaced by the build's own oracle\n"
                "integrator (pins everything except the Bullet integrator/contact: parity unpinned there) -- since round 5 at the\n"
                "model's DEFINITION: contact rows swept to convergence (solver_iters 50), no closed forms for flat bodies (rest_shortcut 0);\n"
                "the library's defaults (cap of 10, closed forms) are held against these files as a stated distance by the tests\n"
                "(rounds 3-4 generated F6 at the then-current defaults and regenerated it with every solver change: 5 files in round 4),\n"
                "F6c the same harness, same settings, on bodies tumbling on the ground,\n"
                "F7 the Reynolds flocking expert of examples/simulating_data (forward_batch, D=6/9, K=1..3),\n"
                "F8 the flocking metrics of examples/simulating_data/helper/MRSAnalytics.py.\n"
                % (np.__version__, scipy.__version__))


if __name__ == "__main__":
    main()
