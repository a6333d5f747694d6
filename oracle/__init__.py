"""ctypes front-end of the CPU oracle (oracle/mrs_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py; the product package (mrs-gym_amd/) never imports it.

Parity status: controller / mixer / force assembly / adjacency / history are pinned
against golden vectors generated from the reference's own Python (tools/gen_golden.py);
the Bullet integrator + contact restatement is PARITY UNPINNED (pybullet is absent).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmrs_oracle.so")

ACT = {
    None: 0,
    "set_speeds": 1,
    "set_control": 2,
    "set_target_accel": 3,
    "set_target_vel": 4,
    "set_target_pos": 5,
    "set_target_ori": 6,
}
ADIM = {0: 0, 1: 4, 2: 4, 3: 3, 4: 3, 5: 3, 6: 3}


class OrcParams(C.Structure):
    _fields_ = [
        ("mass", C.c_double), ("arm", C.c_double), ("kf", C.c_double), ("km", C.c_double),
        ("thrust2weight", C.c_double),
        ("ixx_file", C.c_double), ("iyy_file", C.c_double), ("izz_file", C.c_double),
        ("gnd_eff_coeff", C.c_double), ("prop_radius", C.c_double), ("drag_xy", C.c_double),
        ("drag_z", C.c_double), ("dw1", C.c_double), ("dw2", C.c_double), ("dw3", C.c_double),
        ("prop_x", C.c_double * 4), ("prop_y", C.c_double * 4), ("prop_z", C.c_double * 4),
        ("coll_radius", C.c_double), ("coll_half_len", C.c_double),
        ("gravity", C.c_double), ("dt", C.c_double), ("ctrl_gravity", C.c_double), ("ctrl_dt", C.c_double),
        ("inertia", C.c_double * 3),
        ("lin_damp", C.c_double), ("ang_damp", C.c_double), ("max_coord_vel", C.c_double),
        ("use_gyro", C.c_int),
        ("ground_z", C.c_double), ("friction", C.c_double), ("erp", C.c_double),
        ("contact_threshold", C.c_double),
        ("solver_iters", C.c_int), ("enable_contact", C.c_int), ("pair_contact", C.c_int), ("rest_shortcut", C.c_int),
    ]


class OrcPid(C.Structure):
    _fields_ = [
        ("integral_pos_e", C.c_double * 3), ("d_vel_e", C.c_double * 3),
        ("integral_vel_e", C.c_double * 3), ("integral_ori_e", C.c_double * 3),
        ("last_vel_e", C.c_float * 3), ("last_target_vel", C.c_float * 3),
    ]


PID_DTYPE = np.dtype([
    ("integral_pos_e", "f8", 3), ("d_vel_e", "f8", 3), ("integral_vel_e", "f8", 3),
    ("integral_ori_e", "f8", 3), ("last_vel_e", "f4", 3), ("last_target_vel", "f4", 3),
], align=True)
assert PID_DTYPE.itemsize == C.sizeof(OrcPid)

_lib = None


def build(force=False):
    """Compile oracle/libmrs_oracle.so with gcc (make)."""
    srcs = [os.path.join(_HERE, f) for f in ("mrs_oracle.c", "mrs_sensors.c", "mrs_oracle.h")]
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libmrs_oracle.so"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        dp, fp = C.POINTER(C.c_double), C.POINTER(C.c_float)
        PP, SP = C.POINTER(OrcParams), C.POINTER(OrcPid)
        _lib.orc_params_default.argtypes = [PP]
        _lib.orc_derived.argtypes = [PP, dp]
        _lib.orc_pid_init.argtypes = [C.c_void_p, C.c_int]
        for n in ("orc_euler_to_quat", "orc_quat_to_euler", "orc_quat_to_matrix", "orc_matrix_to_euler_nearest"):
            getattr(_lib, n).argtypes = [dp, dp]
        _lib.orc_observe.argtypes = [dp, dp, dp, dp, fp, fp, fp, fp, fp]
        _lib.orc_observe_batch.argtypes = [C.c_int, dp, dp, dp, dp, fp, fp, fp, fp]
        _lib.orc_adjacency_batch.argtypes = [C.c_int, C.c_int, fp, C.c_double, fp]
        _lib.orc_reynolds.argtypes = [C.c_int, C.c_int, C.c_int, fp, fp]
        _lib.orc_pos_control.argtypes = [PP, SP, fp, fp, fp, fp, fp, dp]
        _lib.orc_vel_control.argtypes = [PP, SP, fp, fp, fp, fp, dp]
        _lib.orc_accel_control.argtypes = [PP, SP, dp, fp, fp, dp]
        _lib.orc_attitude_control.argtypes = [PP, SP, dp, fp, fp, dp, dp]
        _lib.orc_nnls_rpm.argtypes = [PP, C.c_double, C.c_double, C.c_double, C.c_double, dp]
        _lib.orc_nnls_rpm.restype = C.c_int
        _lib.orc_set_control.argtypes = [PP, fp, dp]
        _lib.orc_set_control.restype = C.c_int
        _lib.orc_adjacency.argtypes = [C.c_int, fp, C.c_double, fp]
        _lib.orc_step.argtypes = [PP, C.c_int, C.c_int, dp, dp, dp, dp, C.c_void_p, fp, C.c_int, C.c_int, dp, dp, C.c_int]
        _lib.orc_integrate.argtypes = [PP, dp, dp, dp, dp, dp, dp]
        _lib.orc_step_full.argtypes = [PP, C.c_int, C.c_int, dp, dp, dp, dp, C.c_void_p, fp, C.c_int, C.c_int,
                                       C.c_double, fp, fp, C.c_int]
        ip = C.POINTER(C.c_int)
        _lib.orc_raycast.argtypes = [PP, C.c_int, dp, dp, C.c_int, fp, fp, C.c_int, C.c_int, C.c_float, ip, fp, fp, fp]
        _lib.orc_closest.argtypes = [PP, C.c_int, dp, dp, C.c_int, dp, dp, dp]
        _lib.orc_proximity.argtypes = [PP, C.c_int, dp, dp, dp]
        _lib.orc_contact_rows.argtypes = [PP, dp, dp, dp, dp, C.c_int]
        _lib.orc_contact_solve.argtypes = [PP, dp, dp, dp, dp]
        _lib.orc_spawn_from.argtypes = [C.c_int, C.c_int, fp, C.c_double, fp]
        _lib.orc_spawn_from.restype = C.c_int
    return _lib


def default_params():
    p = OrcParams()
    lib().orc_params_default(C.byref(p))
    return p


def derived(p=None):
    p = p or default_params()
    out = (C.c_double * 7)()
    lib().orc_derived(C.byref(p), out)
    names = ["GravityForce", "HoverRPM", "MaxRPM", "MaxThrust", "MaxXYTorque", "MaxZTorque", "GroundEffectHClip"]
    return dict(zip(names, list(out)))


def _d(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def f64(x, shape=None):
    a = np.ascontiguousarray(np.asarray(x, dtype=np.float64))
    return a.reshape(shape) if shape is not None else a


def f32(x, shape=None):
    a = np.ascontiguousarray(np.asarray(x, dtype=np.float32))
    return a.reshape(shape) if shape is not None else a


def new_pid(n):
    pid = np.zeros(n, dtype=PID_DTYPE)
    lib().orc_pid_init(pid.ctypes.data, n)
    return pid


def euler_to_quat(e):
    e = f64(e); q = np.zeros(4)
    lib().orc_euler_to_quat(_d(e), _d(q))
    return q


def quat_to_euler(q):
    q = f64(q); e = np.zeros(3)
    lib().orc_quat_to_euler(_d(q), _d(e))
    return e


def quat_to_matrix(q):
    q = f64(q); R = np.zeros(9)
    lib().orc_quat_to_matrix(_d(q), _d(R))
    return R.reshape(3, 3)


def matrix_to_euler_nearest(M):
    M = f64(M).reshape(9); e = np.zeros(3)
    lib().orc_matrix_to_euler_nearest(_d(M), _d(e))
    return e


def observe(pos, quat, vel, angvel):
    pos, quat, vel, angvel = f64(pos), f64(quat), f64(vel), f64(angvel)
    o = [np.zeros(3, np.float32) for _ in range(4)] + [np.zeros(9, np.float32)]
    lib().orc_observe(_d(pos), _d(quat), _d(vel), _d(angvel), *[_f(x) for x in o])
    return dict(pos=o[0], euler=o[1], vel=o[2], angvel=o[3], mat=o[4].reshape(3, 3))


class Controller:
    """One QuadControl instance (QuadControl.py:8-127) living in the oracle."""

    def __init__(self, params=None):
        self.p = params or default_params()
        self.pid = new_pid(1)

    def _s(self):
        return C.cast(self.pid.ctypes.data, C.POINTER(OrcPid))

    def pos_control(self, pos, vel, ori, angvel, target_pos):
        a = [f32(x) for x in (pos, vel, ori, angvel, target_pos)]; rpm = np.zeros(4)
        lib().orc_pos_control(C.byref(self.p), self._s(), *[_f(x) for x in a], _d(rpm))
        return rpm

    def vel_control(self, vel, ori, angvel, target_vel):
        a = [f32(x) for x in (vel, ori, angvel, target_vel)]; rpm = np.zeros(4)
        lib().orc_vel_control(C.byref(self.p), self._s(), *[_f(x) for x in a], _d(rpm))
        return rpm

    def accel_control(self, target_accel, ori, angvel):
        ta = f64(target_accel); o, w = f32(ori), f32(angvel); rpm = np.zeros(4)
        lib().orc_accel_control(C.byref(self.p), self._s(), _d(ta), _f(o), _f(w), _d(rpm))
        return rpm

    def attitude_control(self, target_ori, ori, angvel, target_accel=(0., 0., 9.81)):
        to, ta = f64(target_ori), f64(target_accel); o, w = f32(ori), f32(angvel); rpm = np.zeros(4)
        lib().orc_attitude_control(C.byref(self.p), self._s(), _d(to), _f(o), _f(w), _d(ta), _d(rpm))
        return rpm


def nnls_rpm(thrust, tx, ty, tz, params=None):
    p = params or default_params(); rpm = np.zeros(4)
    it = lib().orc_nnls_rpm(C.byref(p), float(thrust), float(tx), float(ty), float(tz), _d(rpm))
    return rpm, it


def set_control(control, params=None):
    p = params or default_params(); rpm = np.zeros(4); c = f32(control)
    lib().orc_set_control(C.byref(p), _f(c), _d(rpm))
    return rpm


def adjacency(pos, comm_range):
    pos = f32(pos); n = pos.shape[0]
    A = np.zeros((n, n), np.float32)
    lib().orc_adjacency(n, _f(pos), float(comm_range), _f(A))
    return A


def reynolds(x_prev):
    """Reynolds.forward_batch of the reference's data generator on x_prev (E,N,D>=6) float32 -> (E,N,3)."""
    x = np.ascontiguousarray(x_prev, dtype=np.float32)
    E, N, D = x.shape
    out = np.zeros((E, N, 3), np.float32)
    lib().orc_reynolds(E, N, D, _f(x), _f(out))
    return out


def raycast(pos, quat, agent, offset, directions, body=True, RANGE=100.0, params=None):
    """Object.raycast of one agent of one env: pos (N,3), quat (N,4) float64; offset / directions (R,3) float32."""
    p = params or default_params()
    pos, quat = f64(pos), f64(quat)
    N = pos.shape[0]
    d = f32(directions).reshape(-1, 3)
    o = np.ascontiguousarray(np.broadcast_to(f32(offset).reshape(-1, 3), d.shape))
    n = d.shape[0]
    obj = np.zeros(n, np.int32)
    pw, pb, dist = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32), np.zeros(n, np.float32)
    lib().orc_raycast(C.byref(p), N, _d(pos), _d(quat), int(agent), _f(o), _f(d), n, int(bool(body)), float(RANGE),
                      obj.ctypes.data_as(C.POINTER(C.c_int)), _f(pw), _f(pb), _f(dist))
    return {"object": obj, "pos world": pw, "pos": pb, "dist": dist}


def closest(pos, quat, agent, params=None):
    """Object.get_dist of one agent against every quadcopter and the ground (index N) of one env."""
    p = params or default_params()
    pos, quat = f64(pos), f64(quat)
    N = pos.shape[0]
    dist, ps, po = np.zeros(N + 1), np.zeros((N + 1, 3)), np.zeros((N + 1, 3))
    lib().orc_closest(C.byref(p), N, _d(pos), _d(quat), int(agent), _d(dist), _d(ps), _d(po))
    return {"distance": dist, "closest pos self": ps, "closest pos other": po}


def proximity(pos, quat, params=None):
    p = params or default_params()
    pos, quat = f64(pos), f64(quat)
    N = pos.shape[0]
    D = np.zeros((N, N + 1))
    lib().orc_proximity(C.byref(p), N, _d(pos), _d(quat), _d(D))
    return D


def contact_rows(params, pos, quat, v, w, n_sweeps):
    """The ground-contact rows of one body swept n_sweeps times without closed forms or early exits: returns (v, w) after."""
    v, w = f64(v).copy(), f64(w).copy()
    lib().orc_contact_rows(C.byref(params), _d(f64(pos)), _d(f64(quat)), _d(v), _d(w), int(n_sweeps))
    return v, w


def contact_solve(params, pos, quat, v, w):
    """The model's ground-contact solve of one body (closed forms as params.rest_shortcut says, the model's stopping rules): (v, w) after."""
    v, w = f64(v).copy(), f64(w).copy()
    lib().orc_contact_solve(C.byref(params), _d(f64(pos)), _d(f64(quat)), _d(v), _d(w))
    return v, w


def spawn_from(cand, agent_radius=0.3):
    """MRS.generate_start_pos on candidate rounds cand (R,N,3) float32 -> (final (N,3) float32, rounds consumed or -1)."""
    c = f32(cand)
    R, N = c.shape[0], c.shape[1]
    pos = np.zeros((N, 3), np.float32)
    used = lib().orc_spawn_from(N, R, _f(c), float(agent_radius), _f(pos))
    return pos, used


def integrate(params, pos, quat, vel, angvel, force_body, torque_body):
    """In-place single-body Bullet step (used by the fake-bullet harness)."""
    lib().orc_integrate(C.byref(params), _d(pos), _d(quat), _d(vel), _d(angvel), _d(f64(force_body)), _d(f64(torque_body)))


class OracleSwarm:
    """E independent envs of N quadcopters stepped by the oracle; arrays are (E,N,k) float64."""

    def __init__(self, n_envs, n_agents, params=None, nthreads=1):
        self.E, self.N = int(n_envs), int(n_agents)
        self.p = params or default_params()
        self.nthreads = nthreads
        self.pos = np.zeros((self.E, self.N, 3))
        self.quat = np.zeros((self.E, self.N, 4)); self.quat[..., 3] = 1.0
        self.vel = np.zeros((self.E, self.N, 3))
        self.angvel = np.zeros((self.E, self.N, 3))
        self.pid = new_pid(self.E * self.N)
        self.speeds = np.zeros((self.E, self.N, 4))
        self.wrench = np.zeros((self.E, self.N, 6))

    def set_state(self, pos=None, euler=None, quat=None, vel=None, angvel=None):
        """Environment.set_state / Object.set_state semantics (None keeps the current value)."""
        if pos is not None:
            self.pos[...] = f64(pos).reshape(self.E, self.N, 3)
        if euler is not None:
            e = f64(np.asarray(euler, dtype=np.float32)).reshape(-1, 3)  # euler arrives as a float32 tensor
            q = np.stack([euler_to_quat(x) for x in e]).reshape(self.E, self.N, 4)
            self.quat[...] = q
        if quat is not None:
            self.quat[...] = f64(quat).reshape(self.E, self.N, 4)
        if vel is not None:
            self.vel[...] = f64(vel).reshape(self.E, self.N, 3)
        if angvel is not None:
            self.angvel[...] = f64(angvel).reshape(self.E, self.N, 3)

    def step(self, actions, action_type):
        at = ACT[action_type] if not isinstance(action_type, int) else action_type
        adim = ADIM[at]
        if actions is None:
            at, a_ptr = 0, None
        else:
            a = f32(actions).reshape(self.E, self.N, adim)
            a_ptr = _f(a)
        lib().orc_step(C.byref(self.p), self.E, self.N, _d(self.pos), _d(self.quat), _d(self.vel), _d(self.angvel),
                       self.pid.ctypes.data, a_ptr, at, adim, _d(self.speeds), _d(self.wrench), int(self.nthreads))

    def step_full(self, actions, action_type, comm_range, want_A=True):
        """CPU baseline: step + newest cat(pos, vel) slice + newest dense adjacency, env-parallel."""
        at = ACT[action_type]
        adim = ADIM[at]
        a = f32(actions).reshape(self.E, self.N, adim)
        if not hasattr(self, "_obs"):
            self._obs = np.zeros((self.E, self.N, 6), np.float32)
            self._A = np.zeros((self.E, self.N, self.N), np.float32)
        lib().orc_step_full(C.byref(self.p), self.E, self.N, _d(self.pos), _d(self.quat), _d(self.vel), _d(self.angvel),
                            self.pid.ctypes.data, _f(a), at, adim, float(comm_range), _f(self._obs),
                            _f(self._A) if want_A else None, int(self.nthreads))
        return self._obs, self._A

    def observe(self):
        """float32 read-back of every agent: dict of (E,N,3) arrays + (E,N,3,3) mat."""
        out = dict(pos=np.zeros((self.E, self.N, 3), np.float32), euler=np.zeros((self.E, self.N, 3), np.float32),
                   vel=np.zeros((self.E, self.N, 3), np.float32), angvel=np.zeros((self.E, self.N, 3), np.float32))
        lib().orc_observe_batch(self.E * self.N, _d(self.pos), _d(self.quat), _d(self.vel), _d(self.angvel),
                                _f(out["pos"]), _f(out["euler"]), _f(out["vel"]), _f(out["angvel"]))
        return out

    def adjacency(self, comm_range):
        p32 = np.ascontiguousarray(self.pos.astype(np.float32))
        A = np.zeros((self.E, self.N, self.N), np.float32)
        lib().orc_adjacency_batch(self.E, self.N, _f(p32), float(comm_range), _f(A))
        return A
