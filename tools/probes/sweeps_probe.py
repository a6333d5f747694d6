"""Diagnostic (MRS_TIMELINE build): how many contact sweeps do grounded bodies need, and how many do their waves run?
In that build the solver writes, per solved body, [first even sweep count at which it had converged | 0] and the number
of sweeps its wave executed into planes 0 / 1 of the controller memory (set_target_pos leaves those planes ... no:
the probe uses ACTION_TYPE None, which touches no controller memory)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'mrs-gym_amd'), os.path.join(ROOT, 'tests')]
import numpy as np, torch, mrsgym_amd
from mrsgym_amd.native import ACT
from util_scenarios import ActionStream, grid_spawn
E, N = 4096, 64
pos, eul = grid_spawn(E, N); z = np.zeros((E, N, 3), np.float32)
sh = mrsgym_amd.SwarmShard(E, N, "cuda:0")
sh.set_state(pos=pos, ori=eul, vel=z, angvel=z)
acts = ActionStream("set_target_vel", E, N, pos, seed=1000)
table = [torch.from_numpy(acts(50 * k)).cuda() for k in range(20)]
obs = torch.zeros(E, N, 6, device="cuda"); adj = torch.zeros(E, N, 1, dtype=torch.int64, device="cuda")
for t in range(800):
    sh.step_ptr(table[(t // 50) % 20], ACT["set_target_vel"], obs.data_ptr(), adj.data_ptr(), 5.0)
torch.cuda.synchronize()
sh.pid[0].zero_()
sh.step_ptr(table[16], ACT["set_target_vel"], obs.data_ptr(), adj.data_ptr(), 5.0)   # planes 0-2 (integral_pos_e) are not used by set_target_vel
torch.cuda.synchronize()
conv = sh.pid[0, :, 0].cpu().numpy(); ran = sh.pid[0, :, 1].cpu().numpy()
solved = ran > 0
print("bodies solved: %d of %d (%.1f %%)" % (solved.sum(), E * N, 100 * solved.mean()))
print("sweeps run by their wave: ", {int(k): int((ran[solved] == k).sum()) for k in np.unique(ran[solved])})
print("first converged at (0 = not within the sweeps run): ", {int(k): int((conv[solved] == k).sum()) for k in np.unique(conv[solved])})
