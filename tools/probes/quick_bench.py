"""Throwaway kernel timing (diagnostic): one mrs_step launch per step, HIP events around the loop."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'mrs-gym_amd'), os.path.join(ROOT, 'tests')]
import numpy as np, torch, mrsgym_amd
from util_scenarios import ActionStream, grid_spawn
E = int(os.environ.get("E", 4096)); N = int(os.environ.get("N", 64)); K = int(os.environ.get("K", 300))
ONLY = os.environ.get("ONLY")
for atype, cr in [("set_target_vel", 5.0), ("set_speeds", float("nan")), ("set_target_pos", 5.0), ("set_control", 5.0), (None, 5.0)]:
    if ONLY and str(atype) != ONLY:
        continue
    pos, eul = grid_spawn(E, N)
    z = np.zeros((E, N, 3), np.float32)
    sh = mrsgym_amd.SwarmShard(E, N, "cuda:0")
    sh.set_state(pos=pos, ori=eul, vel=z, angvel=z)
    acts = ActionStream(atype, E, N, pos, seed=11) if atype else None
    table = [torch.from_numpy(acts(50 * k)).cuda() for k in range(K // 50 + 1)] if atype else None
    obs = torch.zeros(E, N, sh.D, device="cuda:0"); adj = torch.zeros(E, N, sh.W, dtype=torch.int64, device="cuda:0")
    if os.environ.get("NOADJ"): cr = float("nan")
    use_adj = adj if cr == cr else None
    use_obs = None if os.environ.get("NOOBS") else obs
    for t in range(20):
        sh.step(table[0] if table else None, atype, obs_out=use_obs, adj_out=use_adj, comm_range=cr)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for t in range(K):
        sh.step(table[t // 50] if table else None, atype, obs_out=use_obs, adj_out=use_adj, comm_range=cr)
    e1.record(); torch.cuda.synchronize(); t1 = time.perf_counter()
    ms = e0.elapsed_time(e1) / K
    print("%-16s E=%d N=%d  %.1f us/step (wall %.1f us)  %.3g agent-steps/s" % (atype, E, N, ms * 1e3, (t1 - t0) / K * 1e6, E * N / (ms * 1e-3)), flush=True)
