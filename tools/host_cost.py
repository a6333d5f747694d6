"""Diagnostic: pure host cost of one MRS.step() / SwarmShard.step_ptr call -- a swarm small enough (64 envs) that the
kernel (~6 us) is never what the loop waits for, so the loop period IS the host's time per call."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'mrs-gym_amd'), os.path.join(ROOT, 'tests')]
import numpy as np, torch, mrsgym_amd
from mrsgym_amd.native import ACT
from util_scenarios import ActionStream, grid_spawn
E, N = 64, 64
pos, eul = grid_spawn(E, N)
def state_fn(quad):
    return torch.cat([quad.get_pos(), quad.get_vel()])
env = mrsgym_amd.make('mrs-v0', N_ENVS=E, N_AGENTS=N, state_fn=state_fn, K_HOPS=3, COMM_RANGE=5.0, RETURN_A=True, ACTION_TYPE="set_target_vel",
                      HEADLESS=True, START_POS=torch.from_numpy(pos), A_FORMAT="packed", CHECK_NAN="lazy")
env.reset(ori=torch.from_numpy(eul))
a = torch.from_numpy(ActionStream("set_target_vel", E, N, pos, seed=1)(0)).cuda()
for _ in range(3000): env.step(a)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(20000): env.step(a)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("env.step       host %.2f us/call (with final sync %.2f)" % ((t1 - t0) / 20000 * 1e6, (t2 - t0) / 20000 * 1e6), flush=True)
sh = env.shard
obs = torch.zeros(E, N, 6, device="cuda"); adj = torch.zeros(E, N, 1, dtype=torch.int64, device="cuda")
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(20000): sh.step_ptr(a, ACT["set_target_vel"], obs.data_ptr(), adj.data_ptr(), 5.0)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("shard.step_ptr host %.2f us/call (with final sync %.2f)" % ((t1 - t0) / 20000 * 1e6, (t2 - t0) / 20000 * 1e6), flush=True)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(20000): env.step(a)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
