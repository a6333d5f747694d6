"""The step kernel's per-body DEVICE FUNCTIONS (mrs-gym_amd/csrc/mrs_device.hpp), compiled for the CPU, against the oracle.

Not a product path and not a fallback: tools/host_f32 compiles the device header with a stand-in for <hip/hip_runtime.h>
(the handful of gfx950 builtins mapped to libm) into build/libcontact_host.so, and this test drives the rigid-body pipeline of
k_step -- integrate_velocity -> contact_at_rest | contact_solve_f32 -> integrate_pose, the functions themselves -- on random
bodies.  What it adds to the GPU parity tests: the kernel's arithmetic is held against the oracle HERE, without a GPU, so a
change to the device math that breaks parity fails the CPU suite already; and the float64 instantiation of the contact
statements shows kernel and oracle to be the same algorithm (1e-11), the float32 distance being arithmetic.
Skipped where the ROCm clang++ is missing.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
LIB = os.path.join(ROOT, "build", "libcontact_host.so")


@pytest.fixture(scope="module")
def host():
    if not os.path.exists(CLANG):
        pytest.skip("ROCm clang++ not present: the device header cannot be compiled for the host here")
    subprocess.check_call([os.path.join(ROOT, "tools", "host_f32", "build.sh")], stdout=subprocess.DEVNULL)
    import sys
    sys.path.insert(0, os.path.join(ROOT, "mrs-gym_amd"))
    from mrsgym_amd import native              # the ctypes layout of MrsParams only; the HIP library is not loaded
    H = C.CDLL(LIB)
    dp = C.POINTER(C.c_double)
    H.host_body_step.argtypes = [C.POINTER(native.MrsParams), dp, dp, dp, dp, dp, dp]
    for nm in ("host_contact_f64", "host_contact_f32t"):
        getattr(H, nm).argtypes = [C.POINTER(native.MrsParams), C.c_double, dp, dp, dp, dp, dp]
    P = oracle.default_params()
    mp = native.MrsParams()
    for f, _ in native.MrsParams._fields_:
        if hasattr(P, f):
            v = getattr(P, f)
            try:
                setattr(mp, f, v)
            except TypeError:
                for i in range(len(v)):
                    getattr(mp, f)[i] = v[i]
    return H, mp, P


def _d(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _bodies(rng, n, z_lo, z_hi, tilt, speed, flat=False):
    out = []
    for _ in range(n):
        yaw = rng.uniform(-np.pi, np.pi)
        e = np.array([0.0, 0.0, yaw]) if flat else np.array([rng.uniform(-tilt, tilt), rng.uniform(-tilt, tilt), yaw])
        q = np.zeros(4)
        oracle.lib().orc_euler_to_quat(_d(e), _d(q))
        pos = np.array([rng.uniform(-3, 3), rng.uniform(-3, 3), rng.uniform(z_lo, z_hi)])
        v = rng.uniform(-speed, speed, 3)
        w = rng.uniform(-speed, speed, 3) * (0.0 if flat else 1.0)
        if flat:
            v[:2] = 0.0
        fb = np.array([rng.uniform(-0.02, 0.02), rng.uniform(-0.02, 0.02), rng.uniform(0.0, 0.6)])
        tb = rng.uniform(-1e-3, 1e-3, 3) * (0.0 if flat else 1.0)
        out.append((pos, q, v, w, fb, tb))
    return out


def _both(H, mp, P, body):
    pos, q, v, w, fb, tb = body
    a = [x.copy() for x in (pos, q, v, w)]
    H.host_body_step(C.byref(mp), _d(a[0]), _d(a[1]), _d(a[2]), _d(a[3]), _d(fb.copy()), _d(tb.copy()))
    b = [x.copy() for x in (pos, q, v, w)]
    oracle.integrate(P, b[0], b[1], b[2], b[3], fb, tb)
    return np.concatenate(a), np.concatenate(b)


def test_free_flight_body_step_equals_the_oracle(host):
    H, mp, P = host
    rng = np.random.default_rng(5)
    worst = 0.0
    for body in _bodies(rng, 2000, 1.0, 4.0, 1.2, 6.0):
        a, b = _both(H, mp, P, body)
        worst = max(worst, float((np.abs(a - b) / np.maximum(1.0, np.abs(b))).max()))
    # float64 on both sides; the kernel's reciprocals / square roots are ~1 ulp forms, |v| and |w| of the damping term float32
    # square roots (6e-8 relative on a damping term of up to 0.02 m/s per step at these speeds: 1e-9)
    assert worst < 2e-9, worst


def test_ground_contact_body_step_against_the_oracle(host):
    """Bodies within the contact threshold of the ground, velocities of ordinary size: float32 sweeps against float64 sweeps
    (the tolerance of tests/test_gpu_teacher.py for that class, 5e-5); exactly flat bodies take the closed forms on both
    sides and agree to rounding."""
    H, mp, P = host
    rng = np.random.default_rng(6)
    errs = []
    for body in _bodies(rng, 3000, 0.505, 0.56, 0.3, 0.5):
        a, b = _both(H, mp, P, body)
        errs.append(float((np.abs(a - b) / np.maximum(1.0, np.abs(b))).max()))
    errs = np.array(errs)
    assert np.median(errs) < 1e-6 and np.quantile(errs, 0.99) < 5e-5 and errs.max() < 5e-4, (np.median(errs), np.quantile(errs, 0.99), errs.max())
    flat = []
    for body in _bodies(rng, 500, 0.5125 - 2e-4, 0.5125 + 2e-4, 0.0, 0.05, flat=True):
        a, b = _both(H, mp, P, body)
        flat.append(float(np.abs(a - b).max()))
    flat = np.array(flat)     # the few that the closed forms do not take (neither lifting nor sticking) go through the sweeps
    assert (flat < 1e-12).mean() > 0.9 and flat.max() < 1e-6, ((flat < 1e-12).mean(), flat.max())


def test_contact_statements_in_float64_are_the_oracles_algorithm(host):
    """contact_solve_f32's statements with the scalar type switched to double (tools/host_f32/gen_variant.py derives the
    template from the header's text) against the oracle's rows with the model's stopping rules: the same algorithm to rounding."""
    H, mp, P = host
    rng = np.random.default_rng(7)
    Pn = oracle.default_params()
    Pn.rest_shortcut = 0
    worst64 = 0.0
    n = 0
    for pos, q, v, w, fb, tb in _bodies(rng, 1500, 0.505, 0.56, 0.4, 1.0):
        dv, dw = np.zeros(3), np.zeros(3)
        H.host_contact_f64(C.byref(mp), pos[2], _d(q.copy()), _d(v.copy()), _d(w.copy()), _d(dv), _d(dw))
        # the oracle's contact solve alone: rows with the model's early exits, no closed forms
        ref = oracle.contact_solve(Pn, pos, q, v, w)
        worst64 = max(worst64, float(np.abs(v + dv - ref[0]).max()), float(np.abs(w + dw - ref[1]).max()))
        n += 1
    assert n > 0 and worst64 < 1e-10, worst64


MODE_CODE = {"accel": 3, "vel": 4, "pos": 5, "ori": 6}        # MRS_ACT_* (include/mrs_hip.h)


@pytest.mark.parametrize("rounded", [0, 1])
@pytest.mark.parametrize("mode", ["pos", "vel", "accel", "ori"])
def test_F1_reference_controller_through_the_device_functions(host, golden_dir, mode, rounded):
    """The reference's own QuadControl outputs (fixture F1: 256 quadcopters x 5 consecutive calls, a quarter at large attitude)
    against the kernel's controller functions called in k_step's order (tools/host_f32/contact_host.cpp host_controller), both
    controller forms (MrsParams.round_euler_readback).  Tolerance as on the GPU (tests/test_gpu_fixtures.py): 0.05 rpm -- the
    Euler angles reach the controller through quaternion -> float32 -> matrix -> atan2, 1e-7 rad away from the fixture's, times
    the attitude gain."""
    H, mp, P = host
    import copy
    mp2 = copy.copy(mp)
    mp2.round_euler_readback = rounded
    fp = C.POINTER(C.c_float)
    H.host_controller.argtypes = [C.c_void_p, C.c_void_p, C.c_int, fp, fp, fp, fp, fp, C.POINTER(C.c_double)]
    d = np.load(os.path.join(golden_dir, "F1_quadcontrol.npz"))
    m = [str(x) for x in d["modes"]].index(mode)
    n, calls = d["rpm"].shape[1], d["rpm"].shape[2]
    nb = H.host_pid_bytes()
    worst = 0.0
    f = lambda a: np.ascontiguousarray(a, np.float32).ctypes.data_as(fp)
    for i in range(n):
        mem = C.create_string_buffer(nb)
        H.host_pid_init(mem)
        for c in range(calls):
            tgt = d["target"][i, c] * np.float32(0.3) if mode == "ori" else d["target"][i, c]
            rpm = np.zeros(4)
            H.host_controller(C.byref(mp2), mem, MODE_CODE[mode], f(d["pos"][i, c]), f(d["ori"][i, c]), f(d["vel"][i, c]), f(d["angvel"][i, c]),
                              f(tgt), _d(rpm))
            worst = max(worst, float(np.abs(rpm - d["rpm"][m, i, c]).max()))
    assert worst < 0.05, worst


def test_F2_reference_nnls_through_the_device_function(host, golden_dir):
    """nnlsRPM (fixture F2, 43 % of the rows on the NNLS branch) against the kernel's closed-form set_control: rpm^2 within 2e-6
    of the wrench scale of the reference's (the float32 rounding of the control input), as on the GPU."""
    H, mp, P = host
    fp = C.POINTER(C.c_float)
    H.host_set_control.argtypes = [C.c_void_p, fp, C.POINTER(C.c_double)]
    d = np.load(os.path.join(golden_dir, "F2_nnls.npz"))
    k = np.array([P.mass, P.ixx_file, P.iyy_file, P.izz_file])
    ctrl = (d["wrench"] / k).astype(np.float32)
    bc = np.array([1 / P.kf, 1 / (P.kf * P.arm), 1 / (P.kf * P.arm), 1 / P.km])
    w32 = (ctrl * k.astype(np.float32)).astype(np.float64)
    scale = np.abs(w32 * bc).max(1) + 1.0
    got = np.zeros((len(ctrl), 4))
    for i, c in enumerate(ctrl):
        H.host_set_control(C.byref(mp), np.ascontiguousarray(c).ctypes.data_as(fp), _d(got[i]))
    assert d["nnls_branch"].mean() > 0.4
    assert (np.abs(got ** 2 - d["rpm"] ** 2) / scale[:, None]).max() < 2e-6
    orc = np.stack([oracle.set_control(c) for c in ctrl])
    assert (np.abs(got ** 2 - orc ** 2) / scale[:, None]).max() < 1e-9            # closed-form split against Lawson-Hanson on the same input
