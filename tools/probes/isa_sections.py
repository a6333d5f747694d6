"""Diagnostic: static VALU instruction count per phase of k_step<set_target_vel> for the N = 64 instantiation.

    python tools/probes/isa_sections.py [extra -D flags]

Compiles mrs_kernels.hip with -DMRS_MARKS (phase boundaries as '; MRS_MARK k' comments; the generic-N branches of the
run-time-N kernel are in the listing too, so the static counts are upper bounds of what an N = 64 wave executes) to assembly and histograms the vector instructions between consecutive marks, in layout order.
The contact-sweep loop body is counted once (it runs up to solver_iters times)."""
import collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
flags = sys.argv[1:]
out = "/tmp/isa_sections.s"
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-mllvm", "-amdgpu-kernarg-preload-count=14", "-DMRS_MARKS",
                       "-S", "--cuda-device-only", os.path.join(ROOT, "mrs-gym_amd/csrc/mrs_kernels.hip"), "-o", out] + flags,
                      stderr=subprocess.DEVNULL)
txt = open(out).read()
m = re.search(r"^_Z6k_stepILi4ELi256ELb1ELb1EEviiiiPKfPKdS3_S3_S3_8StepArgs:.*?s_endpgm", txt, re.S | re.M)
body = m.group(0).split("\n")
names = {"0": "downwash pairs", "1": "vel/pos control", "20": "read-back + R", "21": "attitude ctrl", "2": "forces/gnd/drag", "22": "integrate vel",
         "3": "stash+ballot", "4": "contact solve", "5": "(after solve)", "6": "pose+store", "7": "obs+adjacency", "8": "end"}
sec, counts, cur = [], collections.OrderedDict(), "start"
counts[cur] = collections.Counter()
for l in body:
    t = l.strip()
    mm = re.match(r"; MRS_MARK (\d+)", t)
    if mm:
        cur = "-> " + names.get(mm.group(1), mm.group(1))
        counts.setdefault(cur, collections.Counter())
        continue
    if not t or t[0] in ";." or t.endswith(":"):
        continue
    op = t.split()[0]
    op = re.sub(r"_e32$|_e64$|_dpp$|_sdwa$", "", op)
    counts[cur][op] += 1
def cls(op):
    if op.startswith("v_"):
        if "f64" in op and not op.startswith("v_cvt"): return "f64"
        if op.startswith("v_cvt"): return "cvt"
        if op in ("v_mov_b32", "v_mov_b64", "v_accvgpr_write_b32", "v_accvgpr_read_b32"): return "mov"
        if op.startswith("v_cndmask"): return "cnd"
        if op.startswith("v_cmp"): return "cmp"
        if op in ("v_rcp_f32", "v_exp_f32", "v_rsq_f32", "v_sqrt_f32", "v_log_f32"): return "trans32"
        if "f32" in op: return "f32"
        return "int/bit"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "scratch_", "buffer_")): return "vmem"
    return "other"
cols = ["f64", "f32", "trans32", "cvt", "mov", "cnd", "cmp", "int/bit", "salu", "lds", "vmem"]
print("%-20s %6s | " % ("section (from mark)", "VALU") + " ".join("%7s" % c for c in cols))
tot = collections.Counter()
for k, c in counts.items():
    agg = collections.Counter()
    for op, n in c.items():
        agg[cls(op)] += n
    valu = sum(agg[x] for x in cols[:8])
    tot.update(agg); tot["VALU"] += valu
    print("%-20s %6d | " % (k[:20], valu) + " ".join("%7d" % agg[x] for x in cols))
print("%-20s %6d | " % ("total (static)", tot["VALU"]) + " ".join("%7d" % tot[x] for x in cols))
