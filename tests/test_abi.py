"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/mrs_hip.h declares; no compute calls (there is no GPU here)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "mrs_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mrs_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    lib = C.CDLL(g.build_hip())
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), "libmrs_hip.so does not export %s" % n
    assert lib.mrs_abi_version() == 5


def test_binding_covers_header():
    from mrsgym_amd import native
    assert sorted(native.EXPORTS) == _declared()


def test_params_struct_layout_and_constants():
    from mrsgym_amd import native
    p = native.default_params()
    # SURVEY.md 8a row P
    assert (p.mass, p.arm, p.kf, p.km) == (0.027, 0.0397, 3.16e-10, 7.94e-12)
    assert (p.gravity, p.dt, p.ctrl_gravity, p.ctrl_dt) == (9.81, 0.01, 9.81, 0.01)
    assert p.ground_z == 0.5 and p.solver_iters == 10 and p.use_gyro == 1 and p.enable_contact == 1
    d = native.derived(p)
    assert d["HoverRPM"] == pytest.approx(14475.809152959684, rel=1e-14)
    assert d["GroundEffectHClip"] == pytest.approx(0.0377637, rel=1e-5)
    # product and oracle restate the same constants independently
    import oracle
    o = oracle.default_params()
    for name, _ in native.MrsParams._fields_:
        if name == "round_euler_readback":   # product-only switch (the oracle always rounds, like the reference)
            continue
        a, b = getattr(p, name), getattr(o, name)
        if hasattr(a, "__len__"):
            assert list(a) == list(b), name
        else:
            assert a == b, name


def test_host_side_argument_errors_need_no_gpu():
    from mrsgym_amd import native
    L = native.lib()
    assert L.mrs_adj_words(64) == 1 and L.mrs_adj_words(65) == 2 and L.mrs_adj_words(256) == 4
    f = (C.c_int32 * 3)(0, 1, 4)
    assert L.mrs_obs_dim(f, 3) == 10
    bad = (C.c_int32 * 1)(99)
    assert L.mrs_obs_dim(bad, 1) < 0
    h = C.c_void_p()
    p = native.default_params()
    rc = L.mrs_create(C.byref(p), 4, 5000, 0, C.byref(h))
    assert rc == -1 and b"1024" in L.mrs_last_error()


def test_no_cpu_fallback():
    import torch
    from mrsgym_amd import native
    with pytest.raises(native.MrsNativeError):
        native.SwarmShard(2, 3, "cpu")
    if not torch.cuda.is_available():
        with pytest.raises(native.MrsNativeError):
            native.SwarmShard(2, 3, "cuda:0")


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "mrs-gym_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", txt, flags=re.M), f
                assert "mrs_oracle" not in txt or "oracle/mrs_oracle.c" in txt, f
