"""Diagnostic: steady-state time per mrs_step against the number of envs (N = 64, the bench workload): how much of the
bench line's step is a fixed latency chain and how much scales with the swarm."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'mrs-gym_amd'), os.path.join(ROOT, 'tests')]
import numpy as np, torch, mrsgym_amd
from mrsgym_amd.native import ACT
from util_scenarios import ActionStream, grid_spawn
N, ROLLIN, K = 64, 700, 300
for E in (256, 512, 1024, 2048, 3072, 4096, 6144, 8192, 12288, 16384):
    pos, eul = grid_spawn(E, N); z = np.zeros((E, N, 3), np.float32)
    sh = mrsgym_amd.SwarmShard(E, N, "cuda:0")
    sh.set_state(pos=pos, ori=eul, vel=z, angvel=z)
    acts = ActionStream("set_target_vel", E, N, pos, seed=1000)
    table = [torch.from_numpy(acts(50 * k)).cuda() for k in range(40)]
    obs = torch.zeros(E, N, 6, device="cuda"); adj = torch.zeros(E, N, 1, dtype=torch.int64, device="cuda")
    t = 0
    for _ in range(ROLLIN):
        sh.step_ptr(table[(t // 50) % 40], ACT["set_target_vel"], obs.data_ptr(), adj.data_ptr(), 5.0); t += 1
    best = 1e9
    for rep in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(K):
            sh.step_ptr(table[(t // 50) % 40], ACT["set_target_vel"], obs.data_ptr(), adj.data_ptr(), 5.0); t += 1
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / K * 1e3)
    print("E = %5d (%5.2f waves per SIMD): %6.2f us per step, %5.2f ns per env, %.3g agent-steps/s, grounded %.2f" % (
        E, E / 1024.0, best, best * 1e3 / E, E * N / best * 1e6, float((sh.pos[2] < 0.6).float().mean())), flush=True)
    del sh
