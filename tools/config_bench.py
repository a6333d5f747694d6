"""Diagnostic: the BASELINE.json configurations other than the bench line, one mrs_step launch per step.

    python tools/config_bench.py            (on the GPU box)

Per configuration: ROLLIN untimed steps (clock ramp + the swarm settles into its steady mix of flying and grounded
bodies), then 3 x K steps between HIP events on the launch stream; prints the best of the three, the grounded share
and agent-steps/s.  Same chaotic inputs as bench.py (tests/util_scenarios.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'mrs-gym_amd'), os.path.join(ROOT, 'tests')]
import numpy as np, torch, mrsgym_amd
from mrsgym_amd.native import ACT
from util_scenarios import ActionStream, grid_spawn

ROLLIN = int(os.environ.get("ROLLIN", 700)); K = int(os.environ.get("K", 300))
CONFIGS = [  # label, E, N, action type, adjacency
    ("C2  N=64 x1024, set_speeds, no A", 1024, 64, "set_speeds", False),
    ("C3  N=64 x4096, set_target_vel, A (bench line)", 4096, 64, "set_target_vel", True),
    ("C4  N=256 x1024, set_control, A", 1024, 256, "set_control", True),
    ("C5/8 N=64 x4096, set_target_pos, A", 4096, 64, "set_target_pos", True),
    # other N: 256-thread workgroups of floor(256/N) envs, 4 resident per CU -- swarm sizes that fill 1024 workgroups exactly,
    # and the same agent count in 1029 / 1041 workgroups (the few extra ones run as a second round: +60 % for the launch)
    ("    N=3 x87040, set_target_vel, A", 87040, 3, "set_target_vel", True),
    ("    N=3 x87381 (1029 workgroups)", 87381, 3, "set_target_vel", True),
    ("    N=12 x21504, set_target_vel, A", 21504, 12, "set_target_vel", True),
    ("    N=12 x21845 (1041 workgroups)", 21845, 12, "set_target_vel", True),
    ("    N=1024 x256, set_target_vel, A (three launches)", 256, 1024, "set_target_vel", True),
]
for label, E, N, atype, want_adj in CONFIGS:
    pos, eul = grid_spawn(E, N)
    z = np.zeros((E, N, 3), np.float32)
    sh = mrsgym_amd.SwarmShard(E, N, "cuda:0")
    sh.set_state(pos=pos, ori=eul, vel=z, angvel=z)
    acts = ActionStream(atype, E, N, pos, seed=1000)
    table = [torch.from_numpy(acts(50 * k)).cuda() for k in range((ROLLIN + 3 * K) // 50 + 2)]
    obs = torch.zeros(E, N, sh.D, device="cuda:0"); adj = torch.zeros(E, N, sh.W, dtype=torch.int64, device="cuda:0")
    at = ACT[atype]; cr = 5.0 if want_adj else float("nan")
    t = 0
    for _ in range(ROLLIN):
        sh.step_ptr(table[t // 50], at, obs.data_ptr(), adj.data_ptr() if want_adj else 0, cr); t += 1
    torch.cuda.synchronize()
    res = []
    for r in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(K):
            sh.step_ptr(table[t // 50], at, obs.data_ptr(), adj.data_ptr() if want_adj else 0, cr); t += 1
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / K * 1e3)
    grounded = float((sh.pos[2] < 0.6).float().mean())
    print("%-52s %7.1f us/step  %.3g agent-steps/s  grounded %.2f" % (label, min(res), E * N / (min(res) * 1e-6), grounded), flush=True)
    del sh, obs, adj, table

# ---- the same configurations through the PRODUCT API (MRS.step), wall clock: what a user's loop sees.  A small swarm's step
# (C2: 12-13 us of kernel) is of the order of the host's cost per call, so that is what this part is about; step_n(S) queues S
# launches per Python call and is the documented loop for E*N <= 65 536 (frame skip / on-device rollouts).
import time
def state_fn(quad):
    return torch.cat([quad.get_pos(), quad.get_vel()])
for label, E, N, atype, want_adj in CONFIGS[:2]:
    pos, eul = grid_spawn(E, N)
    env = mrsgym_amd.make('mrs-v0', N_ENVS=E, N_AGENTS=N, state_fn=state_fn, K_HOPS=3 if want_adj else 0, COMM_RANGE=5.0, RETURN_A=want_adj,
                          ACTION_TYPE=atype, HEADLESS=True, START_POS=torch.from_numpy(pos), A_FORMAT="packed", CHECK_NAN="lazy")
    env.reset(ori=torch.from_numpy(eul))
    acts = ActionStream(atype, E, N, pos, seed=1000)
    table = [torch.from_numpy(acts(50 * k)).cuda() for k in range((ROLLIN + 4 * K * 8) // 50 + 2)]
    t = 0
    for _ in range(ROLLIN):
        env.step(table[t // 50]); t += 1
    torch.cuda.synchronize()
    res, host = [], []
    for r in range(3):
        t0 = time.perf_counter()
        for _ in range(K * 4):
            env.step(table[t // 50]); t += 1
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        res.append((t2 - t0) / (K * 4) * 1e6); host.append((t1 - t0) / (K * 4) * 1e6)
    print("%-52s env.step(): %6.1f us/step wall (host loop %.1f us/call)" % (label, min(res), min(host)), flush=True)
    S = 8
    res = []
    for r in range(3):
        t0 = time.perf_counter()
        for _ in range(K // 2):
            env.step_n(table[t // 50], S); t += S
        torch.cuda.synchronize(); t2 = time.perf_counter()
        res.append((t2 - t0) / (K // 2 * S) * 1e6)
    print("%-52s env.step_n(S=8): %6.1f us/step wall" % (label, min(res)), flush=True)
    del env
