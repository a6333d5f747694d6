"""Diagnostic: the dense-adjacency leg (A_FORMAT="dense") on some boxes of the pool runs at 48-52 us per step instead of 32.
Is it the size of the history ring the 67 MB slices go through (64 slots = 4.3 GB by default)?  Times the leg with several
HISTORY_SLOTS on whatever box this lands on, and a plain 67 MB device-to-device fill for the box's own write rate."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'mrs-gym_amd'), os.path.join(ROOT, 'tests')]
import numpy as np, torch, mrsgym_amd
from util_scenarios import ActionStream, grid_spawn
E, N = 4096, 64
pos, eul = grid_spawn(E, N)
def state_fn(quad):
    return torch.cat([quad.get_pos(), quad.get_vel()])
acts = ActionStream("set_target_vel", E, N, pos, seed=1000)
table = [torch.from_numpy(acts(50 * k)).cuda() for k in range(40)]
x = torch.zeros(64, E, N, N, device="cuda")        # 4.3 GB
for name, f in (("fill one 67 MB slice of a 4.3 GB buffer, rotating", lambda i: x[i % 64].fill_(1.0)), ("fill the same 67 MB slice", lambda i: x[0].fill_(1.0))):
    for i in range(50): f(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(500): f(i)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 500
    print("%-52s %6.1f us  %.2f TB/s" % (name, dt * 1e6, 67.1e6 / dt / 1e12), flush=True)
del x
for fmt, slots in (("packed", 0), ("dense", 0), ("dense", 8), ("dense", 16), ("dense", 256)):
    env = mrsgym_amd.make('mrs-v0', N_ENVS=E, N_AGENTS=N, state_fn=state_fn, K_HOPS=3, COMM_RANGE=5.0, RETURN_A=True, ACTION_TYPE="set_target_vel",
                          HEADLESS=True, START_POS=torch.from_numpy(pos), A_FORMAT=fmt, CHECK_NAN="lazy", HISTORY_SLOTS=slots)
    env.reset(ori=torch.from_numpy(eul))
    t = 0
    for _ in range(700):
        env.step(table[t // 50]); t += 1
    torch.cuda.synchronize(); res = []
    for r in range(3):
        t0 = time.perf_counter()
        for _ in range(300):
            env.step(table[(t // 50) % 40]); t += 1
        torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / 300 * 1e6)
    print("A_FORMAT=%-6s HISTORY_SLOTS=%-3d  %s us per step" % (fmt, slots, " ".join("%.1f" % v for v in res)), flush=True)
    del env
    torch.cuda.empty_cache()
