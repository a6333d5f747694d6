"""Exhaustive check behind mrs_device.hpp:div_ctrl_dt: float32 x / 0.01f as q0 = RN(x r), q = RN(q0 + r (x - d q0)) with
r = RN(1 / d) equals the correctly rounded quotient for EVERY finite float32 x (quotients in the normal range).
float64 stands in for the fused operations: d q0 (24 x 24 bits) and r rem are exact in float64, and rounding a float64
result of +, -, x, / to float32 is innocuous double rounding (53 >= 2 * 24 + 2).

    python tools/probes/divcheck.py [d]          (~1 minute, numpy)"""
import sys, time
import numpy as np
d = np.float32(float(sys.argv[1]) if len(sys.argv) > 1 else 0.01)
r = np.float32(1.0 / np.float64(d))
d64, r64 = np.float64(d), np.float64(r)
bad = total = 0
t0 = time.time()
with np.errstate(all="ignore"):
    for hi in range(1 << 16):                     # all float32 bit patterns, 65 536 at a time
        x = ((np.uint32(hi) << np.uint32(16)) + np.arange(1 << 16, dtype=np.uint32)).view(np.float32)
        x = x[np.isfinite(x)]
        x64 = x.astype(np.float64)
        want = (x64 / d64).astype(np.float32)
        q0 = (x64 * r64).astype(np.float32)
        rem = x64 - d64 * q0.astype(np.float64)
        q1 = (q0.astype(np.float64) + rem * r64).astype(np.float32)
        fin = np.isfinite(want) & (np.abs(want) > 1e-30)
        bad += int((q1[fin] != want[fin]).sum()); total += int(fin.sum())
print("d = %r: checked %d finite float32 x, %d mismatches (%.0f s)" % (float(d), total, bad, time.time() - t0))
