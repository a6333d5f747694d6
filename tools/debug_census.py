"""Diagnostic: how are the grounded bodies spread over envs on the bench workload?  (per-env count of bodies that
the step kernel queues for the contact solve: z - sqrt(r^2 + hl^2) - threshold <= ground)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'mrs-gym_amd'), os.path.join(ROOT, 'tests')]
import numpy as np, torch, mrsgym_amd
from mrsgym_amd.native import ACT
from util_scenarios import ActionStream, grid_spawn
E, N = 4096, 64
pos, eul = grid_spawn(E, N); z = np.zeros((E, N, 3), np.float32)
sh = mrsgym_amd.SwarmShard(E, N, "cuda:0")
sh.set_state(pos=pos, ori=eul, vel=z, angvel=z)
acts = ActionStream("set_target_vel", E, N, pos, seed=1000)
table = [torch.from_numpy(acts(50 * k)).cuda() for k in range(24)]
bound = (0.06 ** 2 + 0.0125 ** 2) ** 0.5 + 0.02 + 0.5
for t in range(1100):
    sh.step_ptr(table[t // 50], ACT["set_target_vel"], 0, 0, float("nan"))
    if t in (99, 299, 499, 699, 899, 1099):
        c = (sh.view(sh.pos)[:, :, 2] <= bound).sum(1).cpu().numpy()
        h = np.bincount(np.minimum(c, 64), minlength=65)
        print("step %4d: grounded %5.1f %%; envs with 0: %4d, 1-4: %4d, 5-16: %4d, 17-32: %4d, 33-63: %4d, all 64: %4d; workgroups (4 envs) with 0: %d"
              % (t + 1, 100 * c.sum() / (E * N), h[0], h[1:5].sum(), h[5:17].sum(), h[17:33].sum(), h[33:64].sum(), h[64], int((c.reshape(-1, 4).sum(1) == 0).sum())), flush=True)
