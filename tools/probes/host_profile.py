"""Diagnostic: where the host time of one MRS.step() goes (cProfile over the bench loop, GPU box)."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'mrs-gym_amd'), os.path.join(ROOT, 'tests')]
import numpy as np, torch, mrsgym_amd
from util_scenarios import ActionStream, grid_spawn
E, N = 4096, 64
pos, eul = grid_spawn(E, N)
def state_fn(quad):
    return torch.cat([quad.get_pos(), quad.get_vel()])
env = mrsgym_amd.make('mrs-v0', N_ENVS=E, N_AGENTS=N, state_fn=state_fn, K_HOPS=3, COMM_RANGE=5.0, RETURN_A=True, ACTION_TYPE="set_target_vel",
                      HEADLESS=True, START_POS=torch.from_numpy(pos), A_FORMAT="packed", CHECK_NAN="lazy")
env.reset(ori=torch.from_numpy(eul))
a = torch.from_numpy(ActionStream("set_target_vel", E, N, pos, seed=1)(0)).cuda()
for _ in range(2000): env.step(a)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20000): env.step(a)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("host loop %.2f us/step, with final sync %.2f us/step" % ((t1 - t0) / 20000 * 1e6, (t2 - t0) / 20000 * 1e6))
pr = cProfile.Profile(); pr.enable()
for _ in range(20000): env.step(a)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
