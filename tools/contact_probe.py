"""k_contact cost vs solver_iters with the whole swarm resting on the ground (diagnostic)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'mrs-gym_amd'), os.path.join(ROOT, 'tests')]
import numpy as np, torch, mrsgym_amd
from util_scenarios import grid_spawn
E, N = 4096, 64
pos, eul = grid_spawn(E, N); pos[..., 2] = 0.5126
z = np.zeros((E, N, 3), np.float32)
for iters in (0, 1, 2, 5, 10):
    p = mrsgym_amd.default_params(); p.solver_iters = iters
    sh = mrsgym_amd.SwarmShard(E, N, "cuda:0", params=p)
    sh.set_state(pos=pos, ori=eul, vel=z, angvel=z)
    for t in range(30): sh.step(None, None)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for t in range(100): sh.step(None, None)
    e1.record(); torch.cuda.synchronize()
    print("solver_iters=%2d: %.1f us/step (k_step<None> + k_contact), max|v| %.2e" % (iters, e0.elapsed_time(e1) * 10, float(sh.vel.abs().max())), flush=True)
