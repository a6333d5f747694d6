"""Host-side cost per step of the product API (diagnostic).

The launch loop is timed WITHOUT a trailing synchronize over a burst short enough that the HIP queue never
fills (the host runs ahead of the GPU), so the figure is host time only."""
import os, sys, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'mrs-gym_amd'), os.path.join(ROOT, 'tests')]
import numpy as np, torch, mrsgym_amd
from util_scenarios import grid_spawn
E, N = int(os.environ.get("E", 4096)), 64
BURST = int(os.environ.get("BURST", 400))
pos, eul = grid_spawn(E, N)
def state_fn(q): return torch.cat([q.get_pos(), q.get_vel()])
env = mrsgym_amd.make('mrs-v0', N_ENVS=E, N_AGENTS=N, state_fn=state_fn, K_HOPS=3, COMM_RANGE=5.0, RETURN_A=True,
                      START_POS=torch.from_numpy(pos), A_FORMAT="packed", CHECK_NAN="lazy", ACTION_TYPE="set_target_vel")
a = torch.zeros(E, N, 3, device="cuda")
for _ in range(200): env.step(a)


def burst(fn, label):
    best = 1e9
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(BURST): fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        best = min(best, (t1 - t0) / BURST * 1e6)
    print("%-28s host %.1f us/step (last burst: + %.1f us/step GPU drain)" % (label, best, (t2 - t1) / BURST * 1e6), flush=True)


sh = env.shard
xr, ar = env._Xring, env._Apacked
burst(lambda: env.step(a), "env.step")
burst(lambda: sh.step_ptr(a, 4, xr.ptr(3), ar.ptr(3), 5.0), "shard.step_ptr")
L, h, pb = sh.L, sh.h, sh._pb_ref
st = torch.cuda.current_stream().cuda_stream
ap = a.data_ptr()
burst(lambda: L.mrs_step(h, pb, ap, 4, sh.obs_codes, sh.n_obs, 5.0, st), "ctypes mrs_step")
burst(lambda: torch.cuda.current_stream(sh.device).cuda_stream, "torch current_stream")
pr = cProfile.Profile(); pr.enable()
for _ in range(BURST): env.step(a)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
