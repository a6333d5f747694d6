"""Diagnostic: the (f)-row kernels on the GPU box -- duration per call, algorithmic bytes per call, achieved GB/s.

    python tools/sensor_bench.py            (under rocprofv3 --kernel-trace --stats for the per-kernel lines of profiles/)

mrs_raycast (k_raycast), mrs_proximity (k_proximity), mrs_reynolds (k_reynolds), mrs_flock_metrics (k_flock_metrics),
mrs_spawn (k_spawn) on swarms of the bench's shape (rolled in 200 steps so that bodies are spread and partly grounded).
Algorithmic bytes: what the call must read and write in 4-byte words (float64 state counted as float32, as SURVEY.md 8d does)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'mrs-gym_amd'), os.path.join(ROOT, 'tests')]
import numpy as np, torch, mrsgym_amd
from mrsgym_amd import native
from util_scenarios import ActionStream, grid_spawn


def timeit(f, n=50):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def swarm(E, N, steps=200):
    pos, eul = grid_spawn(E, N)
    sh = mrsgym_amd.SwarmShard(E, N, "cuda:0")
    z = np.zeros((E, N, 3), np.float32)
    sh.set_state(pos=pos, ori=eul, vel=z, angvel=z)
    acts = ActionStream("set_target_vel", E, N, pos, seed=1000)
    obs = torch.zeros(E, N, 6, device="cuda:0")
    a = None
    for t in range(steps):
        if t % 50 == 0:
            a = torch.from_numpy(acts(t)).cuda()
        sh.step(a, "set_target_vel", obs_out=obs)
    return sh, obs


def line(name, us, nbytes, note=""):
    print("%-44s %9.1f us per call  %8.2f MB algorithmic  %7.1f GB/s  %s" % (name, us, nbytes / 1e6, nbytes / us / 1e3, note), flush=True)


def main():
    E, N = 4096, 64
    sh, obs = swarm(E, N)
    R = 8
    dirs = np.random.default_rng(0).normal(size=(R, 3)).astype(np.float32); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    off = np.array([0, 0, -0.1], np.float32)
    us = timeit(lambda: sh.raycast(off, dirs, body=True, RANGE=5.0), 20)
    line("mrs_raycast  N=64 x4096, 8 rays per agent", us, E * N * (28 + R * 32), "(each ray against 64 cylinders + the ground box)")
    us = timeit(lambda: sh.proximity(points=True), 5)
    line("mrs_proximity N=64 x4096, with points", us, E * N * (28 + (N + 1) * 28), "(4032 GJK pairs per env, float64)")
    us = timeit(lambda: sh.proximity(max_dist=0.02, points=True), 10)
    line("mrs_proximity N=64 x4096, max_dist 0.02, points", us, E * N * (28 + (N + 1) * 28), "(what collisions / get_contact_points ask for)")
    us = timeit(lambda: sh.proximity(max_dist=0.5), 10)
    line("mrs_proximity N=64 x4096, max_dist 0.5", us, E * N * (28 + (N + 1) * 4), "(bounding-sphere cull in front of GJK)")
    act = torch.zeros(E, N, 3, device="cuda:0")
    h = sh.h
    us = timeit(lambda: native._check(sh.L.mrs_reynolds(h, native._ptr(obs), 6, native._ptr(act), native._stream(sh.device)), "mrs_reynolds"), 50)
    line("mrs_reynolds N=64 x4096 (K = 1)", us, E * N * (24 + 12))
    X = obs.reshape(1, E, N, 6).expand(4, E, N, 6).contiguous()
    us = timeit(lambda: native.flock_metrics(X), 20)
    line("mrs_flock_metrics 16384 frames of N=64", us, 4 * E * N * 24 + 4 * E * (N + 4) * 4)
    E2, N2 = 4096, 12
    sh2 = mrsgym_amd.SwarmShard(E2, N2, "cuda:0")
    lo = (native.C.c_float * 3)(0, 0, -1.57); hi = (native.C.c_float * 3)(0, 0, 1.57)
    b = sh2._buffers()
    us = timeit(lambda: native._check(sh2.L.mrs_spawn(sh2.h, native.C.byref(b), 1234, 0, 0.3, lo, hi, 200, None, native._stream(sh2.device)), "mrs_spawn"), 20)
    line("mrs_spawn N=12 x4096 (default distribution)", us, E2 * N2 * 13 * 4, "(rejection rounds in LDS)")


if __name__ == "__main__":
    main()
