"""k_contact cost split on the bench workload (diagnostic): snapshot the C3 swarm at step 600, then time ONE step
from that snapshot with different solver_iters / with the contact pass disabled."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'mrs-gym_amd'), os.path.join(ROOT, 'tests')]
import numpy as np, torch, mrsgym_amd
from mrsgym_amd.native import ACT
from util_scenarios import ActionStream, grid_spawn
E, N = 4096, 64
pos, eul = grid_spawn(E, N)
z = np.zeros((E, N, 3), np.float32)
acts = ActionStream("set_target_vel", E, N, pos, seed=1000)
table = [torch.from_numpy(acts(50 * k)).cuda() for k in range(14)]
at = ACT["set_target_vel"]
sh = mrsgym_amd.SwarmShard(E, N, "cuda:0")
sh.set_state(pos=pos, ori=eul, vel=z, angvel=z)
obs = torch.zeros(E, N, sh.D, device="cuda:0"); adj = torch.zeros(E, N, sh.W, dtype=torch.int64, device="cuda:0")
for t in range(600): sh.step_ptr(table[t // 50], at, obs.data_ptr(), adj.data_ptr(), 5.0)
torch.cuda.synchronize()
snap = {k: v.clone() for k, v in sh.state_dict().items()}
print("bodies within contact range at step 600: %d of %d" % (int((sh.pos[2] < 0.5 + 0.0725 + 0.02).sum()), E * N))
REPS = 30


def one(label, obs_ptr, adj_ptr, **pkw):
    p = mrsgym_amd.default_params()
    for k, v in pkw.items(): setattr(p, k, v)
    sh.set_params(p)
    tot = 0.0
    for r in range(REPS + 3):
        sh.load_state_dict(snap)
        sh.step_ptr(table[12], at, obs_ptr, adj_ptr, 5.0); sh.load_state_dict(snap)   # same parity twice
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); sh.step_ptr(table[12], at, obs_ptr, adj_ptr, 5.0); e1.record(); torch.cuda.synchronize()
        if r >= 3: tot += e0.elapsed_time(e1)
    print("%-44s %.1f us" % (label, tot / REPS * 1e3), flush=True)


one("k_step only (contact off, no obs/adj)", 0, 0, enable_contact=0)
for it in (0, 1, 2, 5, 10):
    one("k_step + k_contact, solver_iters=%d" % it, 0, 0, solver_iters=it)
one("k_step + k_contact + k_observe_adj", obs.data_ptr(), adj.data_ptr())
