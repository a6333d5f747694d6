"""Diagnostic (MRS_TIMELINE build): which grounded bodies need many contact sweeps?  Candidate predictors evaluated on
one step of the steady bench workload: tilt (R22), |w|, |v_xy| of the pre-step state, and the sweep count of the
previous step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'mrs-gym_amd'), os.path.join(ROOT, 'tests')]
import numpy as np, torch, mrsgym_amd
from mrsgym_amd.native import ACT
from util_scenarios import ActionStream, grid_spawn
E, N = 4096, 64
pos, eul = grid_spawn(E, N); z = np.zeros((E, N, 3), np.float32)
sh = mrsgym_amd.SwarmShard(E, N, "cuda:0")
sh.set_state(pos=pos, ori=eul, vel=z, angvel=z)
acts = ActionStream("set_target_vel", E, N, pos, seed=1000)
table = [torch.from_numpy(acts(50 * k)).cuda() for k in range(20)]
obs = torch.zeros(E, N, 6, device="cuda"); adj = torch.zeros(E, N, 1, dtype=torch.int64, device="cuda")
def step(t): sh.step_ptr(table[(t // 50) % 20], ACT["set_target_vel"], obs.data_ptr(), adj.data_ptr(), 5.0)
for t in range(800): step(t)
def probe(t):
    sh.pid[0].zero_(); step(t); torch.cuda.synchronize()
    return sh.pid[0, :, 0].cpu().numpy().copy(), sh.pid[0, :, 1].cpu().numpy().copy()
c0, r0 = probe(800)
q = sh.quat.cpu().numpy(); w = sh.angvel.cpu().numpy(); v = sh.vel.cpu().numpy(); p = sh.pos.cpu().numpy()
c1, r1 = probe(801)
solved = r1 > 0
need = np.where(c1 == 0, 12, c1)          # 12 = not converged within the sweeps
slow = solved & (need > 4)
print("solved %d, slow (>4 sweeps) %d = %.1f %%" % (solved.sum(), slow.sum(), 100 * slow.sum() / solved.sum()))
R22 = 1 - 2 * (q[0] ** 2 + q[1] ** 2); wn = np.sqrt((w ** 2).sum(0)); vxy = np.sqrt(v[0] ** 2 + v[1] ** 2)
prev_slow = (r0 > 0) & (np.where(c0 == 0, 12, c0) > 4)
prev_unsolved = ~(r0 > 0)
def report(name, pred):
    pred = pred & solved
    tp = (pred & slow).sum(); fp = (pred & ~slow).sum(); fn = (~pred & slow).sum()
    print("%-34s predicted %6d  hits %5d  false alarms %6d  missed %5d  (missed per 64 fast bodies: %.2f)" % (name, pred.sum(), tp, fp, fn, 64.0 * fn / max(1, (solved & ~pred).sum())))
report("slow in the previous step", prev_slow)
report("slow or not grounded previously", prev_slow | prev_unsolved)
for thr in (0.9999, 0.999, 0.99): report("tilt R22 < %g" % thr, R22 < thr)
for thr in (0.01, 0.1, 1.0): report("|w| > %g" % thr, wn > thr)
for thr in (0.01, 0.1): report("|v_xy| > %g" % thr, vxy > thr)
report("R22<0.999 or |w|>0.1 or prev", (R22 < 0.999) | (wn > 0.1) | prev_slow | prev_unsolved)
