"""Diagnostic (MRS_TIMELINE build, see tools/probes/timeline_probe.py): how the four waves that share a SIMD progress against each
other, and the waves of one workgroup against each other -- the skew that the workgroup barriers of the contact hand-off
turn into waiting.  Uses the per-wave HW_ID / XCC_ID stamps."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'mrs-gym_amd'), os.path.join(ROOT, 'tests')]
import numpy as np, torch, mrsgym_amd
from mrsgym_amd.native import ACT
from util_scenarios import ActionStream, grid_spawn
E, N = int(os.environ.get("E", 4096)), int(os.environ.get("N", 64))
ATYPE = os.environ.get("ATYPE", "set_target_vel")
pos, eul = grid_spawn(E, N); z = np.zeros((E, N, 3), np.float32)
sh = mrsgym_amd.SwarmShard(E, N, "cuda:0", want_rpm=True)
sh.set_state(pos=pos, ori=eul, vel=z, angvel=z)
acts = ActionStream(ATYPE, E, N, pos, seed=1000)
table = [torch.from_numpy(acts(50 * k)).cuda() for k in range(20)]
obs = torch.zeros(E, N, sh.D, device="cuda"); adj = torch.zeros(E, N, sh.W, dtype=torch.int64, device="cuda")
names = ["loads", "downwash", "controller", "forces", "barrier1", "solve", "barrier2", "pose+store", "obs+adj"]
for t in range(800):
    sh.step_ptr(table[(t // 50) % 20], ACT[ATYPE], obs.data_ptr(), adj.data_ptr(), 5.0)
torch.cuda.synchronize()
WPW = 4                                                   # waves per workgroup (256 threads)
NW = E if N == 64 else -(-E // (256 // N)) * 4            # waves of the launch
full = sh.rpm.flatten()[:NW * 16].view(NW, 16).cpu().numpy().astype(np.float64)
E = NW
hw = full[:, 13].astype(np.int64) | (full[:, 14].astype(np.int64) << 16)
slot, simd, cu, shid, se, xcc = hw & 15, (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7, full[:, 15].astype(np.int64)
t0, t1 = full[:, 11], full[:, 12]
rate = np.median(full[:, 8] / np.maximum((t1 - t0) * 10e-3, 1e-9))     # clock64 ticks per us
print("clock64: %.0f ticks per us; waves %d; distinct (xcc,se,sh,cu): %d, SIMDs: %d" % (rate, E, len(set(zip(xcc, se, shid, cu))), len(set(zip(xcc, se, shid, cu, simd)))))
first = t0.min()
# absolute time (us, chip-wide clock for the start + the wave's own counter for the phases) of each stamp
absT = (t0 - first)[:, None] / 100.0 + full[:, :9] / rate
wg = np.arange(E) // WPW
print("absolute end of each phase, us after the first wave started: mean [p5 .. p95]")
for k, nm in enumerate(names):
    c = absT[:, k]
    print("  %-12s %6.2f [%6.2f .. %6.2f]" % (nm, c.mean(), np.percentile(c, 5), np.percentile(c, 95)))
# skew inside a workgroup and inside a SIMD at the points that matter
def spread(groups, col):
    out = []
    for g in groups:
        c = absT[g, col]
        out.append(c.max() - c.min())
    return np.array(out)
wgs = [np.where(wg == w)[0] for w in range(E // WPW)]
key = xcc * 100000 + se * 10000 + shid * 1000 + cu * 10 + simd
simds = [np.where(key == k)[0] for k in np.unique(key)]
cus = [np.where(key // 10 == k)[0] for k in np.unique(key // 10)]
# where the workgroups' FIRST waves sit (wave 0 runs the contact solve of its workgroup: four of them on one SIMD share its issue slots)
w0 = np.arange(E) % WPW == 0
per_simd = np.array([int(w0[g].sum()) for g in simds])
print("first waves of the workgroups: SIMD id histogram %s; first waves per SIMD: %s (count of SIMDs holding 0, 1, 2, ... of them)" % (
    np.bincount(simd[w0], minlength=4), np.bincount(per_simd)))
for j in range(WPW):
    print("   wave %d of its workgroup sits on SIMD: %s" % (j, np.bincount(simd[np.arange(E) % WPW == j], minlength=4)))
print("waves per SIMD: min %d max %d; per CU: min %d max %d" % (min(map(len, simds)), max(map(len, simds)), min(map(len, cus)), max(map(len, cus))))
for k in (1, 2, 3, 5, 8):
    print("  spread of '%s' end: within a workgroup %.2f us (mean), within a SIMD %.2f, within a CU %.2f, whole grid %.2f" % (
        names[k], spread(wgs, k).mean(), spread(simds, k).mean(), spread(cus, k).mean(), absT[:, k].max() - absT[:, k].min()))
# the two workgroups of a CU: end times
ends = []
for c in cus:
    w = np.unique(wg[c])
    ends.append(sorted(absT[wg == x, 8].max() for x in w))
ne = np.array([len(x) for x in ends])
print("workgroups per CU: ", np.bincount(ne))
cu_end = np.array([x[-1] for x in ends])
print("CU done: min %.2f  p25 %.2f  median %.2f  p75 %.2f  p95 %.2f  max %.2f us" % (cu_end.min(), *np.percentile(cu_end, [25, 50, 75, 95]), cu_end.max()))
wend = absT[:, 8]
print("wave ends: " + "  ".join("p%d %.1f" % (q, np.percentile(wend, q)) for q in (5, 25, 50, 75, 90, 95, 99, 100)))
late = wend > np.percentile(wend, 95)
print("the latest 5%% of the waves: phase durations (us) against everybody's:")
dur = np.diff(np.concatenate([np.zeros((E, 1)), full[:, :9] / rate], 1), axis=1)
for k, nm in enumerate(names):
    print("    %-12s late %.2f   all %.2f" % (nm, dur[late, k].mean(), dur[:, k].mean()))
wg_end = np.array([absT[wg == x, 8].max() for x in range(E // WPW)])
order = np.argsort(-wg_end)[:12]
print("the latest workgroups: end (us), then per wave the phase durations " + "/".join(n[:5] for n in names))
pf = None
zz = sh.pos[2].view(-1, N).cpu().numpy()
low = (zz < 0.6).sum(1)                                  # bodies near the ground per env (after the step)
for x in order:
    ws = np.where(wg == x)[0]
    print("  wg %4d end %.2f  xcc %d cu %d" % (x, wg_end[x], xcc[ws[0]], cu[ws[0]]) + ("  near the ground per env: %s" % low[ws] if N == 64 else ""))
    for w_ in ws:
        print("      " + " ".join("%5.2f" % d for d in dur[w_]) + "   start %.2f" % ((t0[w_] - first) / 100))
widx = np.arange(E) % WPW
print("wave index in its workgroup x SIMD (number of waves):")
for k in range(WPW):
    print("    wave %d: " % k + " ".join("%5d" % ((widx == k) & (simd == j)).sum() for j in range(4)))
print("phase durations by wave index (us): " + "/".join(n[:5] for n in names))
for k in range(WPW):
    print("    wave %d: " % k + " ".join("%5.2f" % dur[widx == k, c].mean() for c in range(9)))
print("phase durations by SIMD (us):")
for j in range(4):
    print("    simd %d: " % j + " ".join("%5.2f" % dur[simd == j, c].mean() for c in range(9)))
print("   late waves per xcc:", np.bincount(xcc[late], minlength=8), " wave index in workgroup:", np.bincount(np.arange(E)[late] % WPW, minlength=WPW))
two = np.array([x for x in ends if len(x) == 2])
if len(two):
    print("CUs with two workgroups: first ends %.2f us, second %.2f us (mean); last CU done at %.2f" % (two[:, 0].mean(), two[:, 1].mean(), two[:, 1].max()))
# per-SIMD: order of the four waves by wave slot
for col in (1, 3, 8):
    rows = []
    for g in simds:
        if len(g) != 4: continue
        o = g[np.argsort(slot[g])]
        rows.append(absT[o, col] - absT[o, col].min())
    rows = np.array(rows)
    print("  '%s' end relative to the SIMD's first, by wave slot order: %s" % (names[col], np.round(rows.mean(0), 2)))
    # by workgroup age on the SIMD
# per XCD mean end
for x in np.unique(xcc):
    m = xcc == x
    print("  xcc %d: waves %d, start %.2f, end mean %.2f max %.2f" % (x, m.sum(), (t0[m] - first).mean() / 100, absT[m, 8].mean(), absT[m, 8].max()))
