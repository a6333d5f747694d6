"""Host-side cost per step of the product API (diagnostic): tiny swarm so the GPU is never the limit."""
import os, sys, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'mrs-gym_amd'), os.path.join(ROOT, 'tests')]
import numpy as np, torch, mrsgym_amd
from util_scenarios import grid_spawn
E, N = 8, 64
pos, eul = grid_spawn(E, N)
def state_fn(q): return torch.cat([q.get_pos(), q.get_vel()])
env = mrsgym_amd.make('mrs-v0', N_ENVS=E, N_AGENTS=N, state_fn=state_fn, K_HOPS=3, COMM_RANGE=5.0, RETURN_A=True,
                      START_POS=torch.from_numpy(pos), A_FORMAT="packed", CHECK_NAN="lazy")
a = torch.zeros(E, N, 3, device="cuda")
for _ in range(200): env.step(a)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2000): env.step(a)
t1 = time.perf_counter(); torch.cuda.synchronize()
print("env.step host cost: %.1f us/step" % ((t1 - t0) / 2000 * 1e6))
sh = env.shard
xr, ar = env._Xring, env._Apacked
t0 = time.perf_counter()
for _ in range(2000): sh.step_ptr(a, 4, xr.ptr(3), ar.ptr(3), 5.0)
t1 = time.perf_counter(); torch.cuda.synchronize()
print("shard.step_ptr host cost: %.1f us/step" % ((t1 - t0) / 2000 * 1e6))
pr = cProfile.Profile(); pr.enable()
for _ in range(2000): env.step(a)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
