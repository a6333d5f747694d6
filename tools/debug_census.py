"""Census of the bench workload's swarm state over time (diagnostic)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'mrs-gym_amd'), os.path.join(ROOT, 'tests')]
import numpy as np, torch, mrsgym_amd
from util_scenarios import ActionStream, grid_spawn
E, N = 1024, 64
pos, eul = grid_spawn(E, N)
z = np.zeros((E, N, 3), np.float32)
sh = mrsgym_amd.SwarmShard(E, N, "cuda:0")
sh.set_state(pos=pos, ori=eul, vel=z, angvel=z)
acts = ActionStream("set_target_vel", E, N, pos, seed=1000)
a = None
for t in range(1100):
    if t % 50 == 0:
        a = torch.from_numpy(acts(t)).cuda()
    sh.step(a, "set_target_vel")
    if t % 100 == 99:
        p = sh.view(sh.pos); v = sh.view(sh.vel)
        bad = ~torch.isfinite(p).all(-1)
        low = (p[..., 2] < 0.6) & ~bad
        fast = (v.norm(dim=-1) > 5) & ~bad
        print(t, "non-finite %.4f  grounded(z<0.6) %.3f  |v|>5 %.3f  max|p| %.1f" % (
            bad.float().mean(), low.float().mean(), fast.float().mean(), float(p[~bad].abs().max())), flush=True)
