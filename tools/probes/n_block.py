"""Diagnostic: step time of one (E, N, action type) configuration; MRS_STEP_BLOCK / MRS_HIP_LIB select the build."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'mrs-gym_amd'), os.path.join(ROOT, 'tests')]
import numpy as np, torch, mrsgym_amd
from mrsgym_amd.native import ACT
from util_scenarios import ActionStream, grid_spawn
E, N, atype = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
pos, eul = grid_spawn(E, N); z = np.zeros((E, N, 3), np.float32)
sh = mrsgym_amd.SwarmShard(E, N, "cuda:0")
sh.set_state(pos=pos, ori=eul, vel=z, angvel=z)
acts = ActionStream(atype, E, N, pos, seed=1000)
table = [torch.from_numpy(acts(50 * k)).cuda() for k in range(40)]
obs = torch.zeros(E, N, sh.D, device="cuda:0"); adj = torch.zeros(E, N, sh.W, dtype=torch.int64, device="cuda:0")
t = 0
for _ in range(700):
    sh.step_ptr(table[t // 50], ACT[atype], obs.data_ptr(), adj.data_ptr(), 5.0); t += 1
torch.cuda.synchronize()
res = []
for r in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(300):
        sh.step_ptr(table[t // 50], ACT[atype], obs.data_ptr(), adj.data_ptr(), 5.0); t += 1
    e1.record(); torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1) / 300 * 1e3)
print("E=%d N=%d %s block=%s: %.1f us/step  checksum %.10e" % (E, N, atype, os.environ.get("MRS_STEP_BLOCK", "default"), min(res), float(sh.pos.double().abs().sum())), flush=True)
