"""Diagnostic: where does bench.py's wall time per step go beyond the three kernels?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'mrs-gym_amd'), os.path.join(ROOT, 'tests')]
import numpy as np, torch, mrsgym_amd
from util_scenarios import ActionStream, grid_spawn
E, N, K = 4096, 64, 1000
pos, eul = grid_spawn(E, N)
def state_fn(q): return torch.cat([q.get_pos(), q.get_vel()])
acts = ActionStream("set_target_vel", E, N, pos, seed=1000)
table = [torch.from_numpy(acts(50 * k)).cuda() for k in range((K + 100) // 50 + 1)]


def run(label, check_nan, direct=False, slots=None):
    kw = dict(HISTORY_SLOTS=slots) if slots else {}
    env = mrsgym_amd.make('mrs-v0', N_ENVS=E, N_AGENTS=N, state_fn=state_fn, K_HOPS=3, COMM_RANGE=5.0, RETURN_A=True,
                          START_POS=torch.from_numpy(pos), A_FORMAT="packed", CHECK_NAN=check_nan,
                          ACTION_TYPE="set_target_vel", **kw)
    env.reset(ori=torch.from_numpy(eul))
    sh, xr, ar = env.shard, env._Xring, env._Apacked
    def one(t):
        if direct:
            sh.step_ptr(table[t // 50], 4, xr.ptr(t % 4), ar.ptr(t % 4), 5.0)
        else:
            env.step(table[t // 50])
    for t in range(100): one(t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(100, 100 + K): one(t)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%-44s wall %.1f us/step (host loop %.1f)" % (label, (t2 - t0) / K * 1e6, (t1 - t0) / K * 1e6), flush=True)


run("env.step CHECK_NAN=lazy", "lazy")
run("env.step CHECK_NAN=off", "off")
run("shard.step_ptr on the env's rings", "off", direct=True)
run("env.step CHECK_NAN=lazy (again)", "lazy")
