"""Diagnostic (CPU): replay the worst teacher-forced cases that `tools/teacher_probe.py --dump` wrote on the GPU box.
For every case (a body's pre-step state, the wrench the oracle applied, the GPU's and the oracle's post-step state): step the body
on its own with the oracle's contact solve forced to exactly k sweeps (k = 0 ... 14, 50) and report which k the GPU's result is
closest to, and how close -- a result that matches another sweep count is a stopping decision that fell differently in float32;
one that matches none is arithmetic."""
import ctypes as C
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import oracle

d = np.load(sys.argv[1])
L = oracle.lib()
L.orc_debug_force_sweeps.argtypes = [C.c_int]
P = oracle.default_params()
KS = list(range(0, 16, 2))[1:] + [50]


def step(pre, wrench, k):
    L.orc_debug_force_sweeps(k)
    pos, quat, vel, ang = pre[0:3].copy(), pre[3:7].copy(), pre[7:10].copy(), pre[10:13].copy()
    oracle.integrate(P, pos, quat, vel, ang, wrench[:3], wrench[3:])
    L.orc_debug_force_sweeps(0)
    return np.concatenate([pos, quat, vel, ang])


def err(a, b):
    return (np.abs(a - b) / np.maximum(1.0, np.abs(b))).max()


n = len(d["err"])
print("%d cases; phases %s" % (n, np.bincount(d["phase"])))
rows = []
for i in range(n):
    if d["phase"][i] == 3:
        continue                     # quad-quad contact needs the partner: not replayed here
    pre, w, gpu, orc = d["pre"][i], d["wrench"][i], d["gpu"][i], d["orc"][i]
    own = step(pre, w, 0)
    e_own = err(own, orc)            # the replay reproduces the dumped oracle result (0 sweeps forced = the model's rule)
    res = {k: step(pre, w, k) for k in KS}
    eg = {k: err(gpu, res[k]) for k in KS}
    eo = {k: err(orc, res[k]) for k in KS}
    kg = min(eg, key=eg.get); ko = min(eo, key=eo.get)
    rows.append((d["err"][i], e_own, ko, eo[ko], kg, eg[kg], err(res[10], res[50]), i))
rows.sort(reverse=True)
print("   err(gpu,orc)  replay==orc  oracle stopped at k (dist)   gpu closest to k (dist)   |x10-x50|   word of the largest difference")
names = ["px", "py", "pz", "qx", "qy", "qz", "qw", "vx", "vy", "vz", "wx", "wy", "wz"]
for r in rows[:40]:
    i = r[7]
    j = int(np.argmax(np.abs(d["gpu"][i] - d["orc"][i])))
    print("   %.2e     %.1e      k=%2d (%.1e)              k=%2d (%.1e)          %.1e     %s   pre: z %.4f tilt %.2e |w| %.2e |v| %.2e"
          % (r[0], r[1], r[2], r[3], r[4], r[5], r[6], names[j], d["pre"][i][2],
             np.hypot(2 * (d["pre"][i][3] * d["pre"][i][5] - d["pre"][i][6] * d["pre"][i][4]), 2 * (d["pre"][i][4] * d["pre"][i][5] + d["pre"][i][6] * d["pre"][i][3])),
             np.abs(d["pre"][i][10:13]).max(), np.abs(d["pre"][i][7:10]).max()))
same = sum(1 for r in rows if r[2] == r[4]); expl = sum(1 for r in rows if r[5] < 0.1 * r[0])
print("%d replayed; gpu closest to the oracle's own sweep count: %d; gpu explained (10x closer) by some sweep count: %d" % (len(rows), same, expl))
