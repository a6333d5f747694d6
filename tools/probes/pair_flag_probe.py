import os, sys
sys.path[:0] = ['/root/repo', '/root/repo/mrs-gym_amd', '/root/repo/tests']
import numpy as np, torch, mrsgym_amd
from mrsgym_amd.native import ACT
from util_scenarios import ActionStream, grid_spawn
E, N = 4096, 64
pos, eul = grid_spawn(E, N); z = np.zeros((E, N, 3), np.float32)
sh = mrsgym_amd.SwarmShard(E, N, "cuda:0")
sh.set_state(pos=pos, ori=eul, vel=z, angvel=z)
acts = ActionStream("set_target_vel", E, N, pos, seed=1000)
table = [torch.from_numpy(acts(50 * k)).cuda() for k in range(60)]
obs = torch.zeros(E, N, 6, device="cuda"); adj = torch.zeros(E, N, 1, dtype=torch.int64, device="cuda")
for t in range(2400):
    sh.step_ptr(table[t // 50], ACT["set_target_vel"], obs.data_ptr(), adj.data_ptr(), 5.0)
    if t % 300 == 299:
        p = sh.pos.view(3, E, N).permute(1, 2, 0).float()
        d = torch.cdist(p, p) + 10 * torch.eye(N, device="cuda")
        close = (d < 0.14)
        print("step %4d: envs with a pair within 0.14 m: %.3f   agents in such a pair: %.4f   grounded %.3f" % (t + 1, float(close.any(-1).any(-1).float().mean()), float(close.any(-1).float().mean()), float((sh.pos[2] < 0.6).float().mean())), flush=True)
