"""mrs_device.hpp:div_ctrl_dt -- float32 division by the controller's DT as multiply + two fused multiply-adds -- is the
correctly rounded quotient (QuadControl.py:62 divides float32 by DT).  tools/probes/divcheck.py is the exhaustive form (all 3.8e9
finite x, a minute); here: 4M random bit patterns plus the neighbourhoods of powers of two, for the default DT and two more."""
import numpy as np
import pytest


@pytest.mark.parametrize("dt", [0.01, 0.005, 0.02])
def test_fma_division_by_ctrl_dt_is_correctly_rounded(dt):
    d = np.float32(dt)
    r = np.float32(1.0 / np.float64(d))
    assert (np.frombuffer(d.tobytes(), np.uint32)[0] & 0x7FFFFF) != 0x7FFFFF      # the kernel's ctrl_div_fast condition
    rng = np.random.default_rng(7)
    bits = rng.integers(0, 1 << 32, 1 << 22, dtype=np.uint64).astype(np.uint32)
    edge = (np.arange(1, 254, dtype=np.uint32)[:, None] << np.uint32(23)) + np.arange(-64, 64, dtype=np.int64)[None, :].astype(np.uint32)
    x = np.concatenate([bits, edge.ravel(), edge.ravel() | np.uint32(1 << 31)]).view(np.float32)
    x = x[np.isfinite(x)]
    x64, d64, r64 = x.astype(np.float64), np.float64(d), np.float64(r)
    with np.errstate(all="ignore"):
        want = (x64 / d64).astype(np.float32)
        q0 = (x64 * r64).astype(np.float32)
        q1 = (q0.astype(np.float64) + (x64 - d64 * q0.astype(np.float64)) * r64).astype(np.float32)
    ok = np.isfinite(want) & (np.abs(want) > 1e-30)
    assert np.array_equal(q1[ok], want[ok])
    # outside the refinement's domain (round 5, MRS_DIV_GUARD 2: a select, no branch) the kernel keeps the first quotient q0 wherever q0
    # is not a normal number (v_cmp_class_f32): q0 IS the division's result for +-inf, for an overflowing product and for +-0; a
    # subnormal q0 may be one subnormal ulp (1.4e-45) off the correctly rounded quotient.  Inside the guard the refinement must be
    # exact -- checked above for |want| > 1e-30 -- and everything outside it must be caught.
    big = np.array([np.inf, -np.inf, 3.4e38, -3.4e38, 3.3e36, 2.9e36, 1e-40, -1e-40, 1.3e-40, 3e-43, 0.0, -0.0], np.float32)
    with np.errstate(all="ignore"):
        q0b = (big.astype(np.float64) * r64).astype(np.float32)
        guarded = ~(np.isfinite(q0b) & (np.abs(q0b) >= np.finfo(np.float32).tiny))
        wantb = (big.astype(np.float64) / d64).astype(np.float32)
        q1b = (q0b.astype(np.float64) + (big.astype(np.float64) - d64 * q0b.astype(np.float64)) * r64).astype(np.float32)
    assert guarded[0] and guarded[1] and guarded[-1] and guarded[-2]        # inf, -inf, 0, -0
    assert np.isnan(q1b[0]) and np.isinf(wantb[0])                          # what the unguarded form would have returned
    un = ~guarded
    assert np.array_equal(q1b[un], wantb[un])
    got = np.where(guarded, q0b, q1b)                                       # the kernel's select
    sub = guarded & np.isfinite(q0b) & (q0b != 0)
    exact = guarded & ~sub
    assert np.array_equal(got[exact], wantb[exact]) and np.array_equal(np.signbit(got[exact]), np.signbit(wantb[exact]))
    ulp = np.abs(got[sub].view(np.int32).astype(np.int64) - wantb[sub].view(np.int32).astype(np.int64))
    assert sub.any() and ulp.max() <= 1
