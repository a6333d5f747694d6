"""Would env-chunks on separate HIP streams overlap the three latency-bound kernels?  (diagnostic)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'mrs-gym_amd'), os.path.join(ROOT, 'tests')]
import numpy as np, torch, mrsgym_amd
from util_scenarios import ActionStream, grid_spawn
N, K = 64, 600
def build(E, base):
    pos, eul = grid_spawn(E, N, env_base=base); z = np.zeros((E, N, 3), np.float32)
    sh = mrsgym_amd.SwarmShard(E, N, "cuda:0"); sh.set_state(pos=pos, ori=eul, vel=z, angvel=z)
    acts = ActionStream("set_target_vel", E, N, pos, seed=base)
    tab = [torch.from_numpy(acts(50 * k)).cuda() for k in range(K // 50 + 1)]
    obs = torch.zeros(E, N, 6, device="cuda"); adj = torch.zeros(E, N, 1, dtype=torch.int64, device="cuda")
    return sh, tab, obs, adj
for chunks in (1, 2, 4):
    E = 4096 // chunks
    parts = [build(E, c * E) for c in range(chunks)]
    streams = [torch.cuda.Stream() for _ in range(chunks)]
    def run(n):
        for t in range(n):
            for (sh, tab, obs, adj), st in zip(parts, streams):
                with torch.cuda.stream(st):
                    sh.step_ptr(tab[t // 50], 4, obs.data_ptr(), adj.data_ptr(), 5.0)
    run(300); torch.cuda.synchronize()      # into the chaotic regime
    t0 = time.perf_counter(); run(K - 300 if K > 300 else K); torch.cuda.synchronize(); t1 = time.perf_counter()
    print("chunks=%d on %d streams: %.1f us per whole step" % (chunks, chunks, (t1 - t0) / (K - 300) * 1e6), flush=True)
