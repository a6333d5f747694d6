"""Diagnostic (CPU): the kernel's float32 contact solve compiled for the host (tools/host_f32/contact_host.cpp) on the dumped
worst cases of tools/teacher_probe.py --dump: does it reproduce the GPU's result, and the GPU's distance to the oracle?"""
import ctypes as C
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mrs-gym_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import oracle
from mrsgym_amd import native

H = C.CDLL(os.path.join(ROOT, "build", os.environ.get("HOSTLIB", "libcontact_host.so")))
dp, fp = C.POINTER(C.c_double), C.POINTER(C.c_float)
H.host_contact.argtypes = [C.POINTER(native.MrsParams), C.c_double, dp, dp, dp, fp, fp]
VARS = ("host_contact_f32t", "host_contact_f64")
for nm in VARS:
    getattr(H, nm).argtypes = [C.POINTER(native.MrsParams), C.c_double, dp, dp, dp, dp, dp]
D = lambda a: a.ctypes.data_as(dp)
F = lambda a: a.ctypes.data_as(fp)
d = np.load(sys.argv[1])
P = oracle.default_params()
Pn = native.default_params() if False else None


def mrs_params():
    """MrsParams filled from the oracle's defaults (same field names), without loading the HIP library."""
    p = native.MrsParams()
    for f, _ in native.MrsParams._fields_:
        if hasattr(P, f):
            v = getattr(P, f)
            try:
                setattr(p, f, v)
            except TypeError:
                for i in range(len(v)):
                    getattr(p, f)[i] = v[i]
    return p


MP = mrs_params()
rows = []
for i in range(len(d["err"])):
    if d["phase"][i] == 3:
        continue
    pre, wr, gpu, orc = d["pre"][i], d["wrench"][i], d["gpu"][i], d["orc"][i]
    P.enable_contact = 0
    pos, quat, v, w = pre[0:3].copy(), pre[3:7].copy(), pre[7:10].copy(), pre[10:13].copy()
    oracle.integrate(P, pos, quat, v, w, wr[:3], wr[3:])
    P.enable_contact = 1
    dv, dw = np.zeros(3, np.float32), np.zeros(3, np.float32)
    H.host_contact(C.byref(MP), pre[2], D(pre[3:7].copy()), D(v), D(w), F(dv), F(dw))
    vh, wh = v + dv, w + dw
    e_go = max(np.abs(gpu[7:10] - orc[7:10]).max(), np.abs(gpu[10:13] - orc[10:13]).max())
    e_hg = max(np.abs(vh - gpu[7:10]).max(), np.abs(wh - gpu[10:13]).max())
    e_ho = max(np.abs(vh - orc[7:10]).max(), np.abs(wh - orc[10:13]).max())
    ex = []
    for nm in VARS:
        a, b = np.zeros(3), np.zeros(3)
        getattr(H, nm)(C.byref(MP), pre[2], D(pre[3:7].copy()), D(v), D(w), D(a), D(b))
        ex.append(max(np.abs(v + a - orc[7:10]).max(), np.abs(w + b - orc[10:13]).max()))
    rows.append((e_go, e_hg, e_ho, i) + tuple(ex))
rows.sort(reverse=True)
print("  |gpu-oracle|  |host-gpu|  |host-oracle|   distance to the oracle of: template<float>  template<double>  split 32/32  set-up 64 + sweeps 32  set-up 32 + sweeps 64")
for r in rows[:25]:
    print("   %.2e     %.2e    %.2e    " % r[:3] + "    ".join("%.2e" % x for x in r[4:]) + "    case %d  z %.4f" % (r[3], d["pre"][r[3]][2]))
