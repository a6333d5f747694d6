"""profiles/traffic.json + trimmed counter CSVs from a tools/profile_round.sh run.

usage: python tools/make_traffic.py gpurun_out/prof_<tag> <name> [commit]     (writes profiles/<name>_*.csv, profiles/traffic.json)"""
import csv, json, os, shutil, sys, collections
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, name = sys.argv[1], sys.argv[2]
commit = sys.argv[3] if len(sys.argv) > 3 else "?"
ALGO = 268 * 4096 * 64


def ours(k):
    # our kernels, without the device warm-up of tools/profile_round.sh (a scratch swarm on k_step<5, ...>: tens of thousands of rows)
    return k.startswith(("k_", "void k_")) and not k.replace("void ", "").startswith("k_step<5")


raw = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = list(csv.DictReader(open(os.path.join(src, c, "run_counter_collection.csv"))))
    keep = [r for r in rows if ours(r["Kernel_Name"])]
    with open(os.path.join(ROOT, "profiles", "%s_pmc_%s.csv" % (name, c)), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader(); w.writerows(keep)
    per = collections.defaultdict(list)
    for r in keep:
        per[r["Kernel_Name"].replace("void ", "").split("(")[0]].append(float(r["Counter_Value"]))
    raw[c] = {k: {"mean_KB_raw": float(np.mean(v)), "launches": len(v)} for k, v in per.items()}
step = [k for k in raw["FETCH_SIZE"] if k.startswith("k_step<4")][0]
fetch = 2 * raw["FETCH_SIZE"][step]["mean_KB_raw"] * 1024
write = raw["WRITE_SIZE"][step]["mean_KB_raw"] * 1024
# SQ pass: VALU instructions per wave and the share of the kernel's cycles in which the VALU was executing
sq_rows = [r for r in csv.DictReader(open(os.path.join(src, "SQ", "run_counter_collection.csv"))) if ours(r["Kernel_Name"])]
with open(os.path.join(ROOT, "profiles", "%s_pmc_SQ.csv" % name), "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(sq_rows[0].keys()))
    w.writeheader(); w.writerows(sq_rows)
sq = collections.defaultdict(list)
for r in sq_rows:
    if r["Kernel_Name"].replace("void ", "").split("(")[0] == step:
        sq[r["Counter_Name"]].append(float(r["Counter_Value"]))
sq = {k: float(np.mean(v)) for k, v in sq.items()}
# SQ_ACTIVE_INST_VALU counts quad-cycles summed over the 1024 SIMDs; SQ_BUSY_CYCLES is summed over the 32 shader engines
valu_busy = (sq["SQ_ACTIVE_INST_VALU"] * 4 / 1024) / (sq["SQ_BUSY_CYCLES"] / 32)
out = {
    "round": 5,
    "source": "profiles/%s_* (commit %s)" % (name, commit),
    "command": "tools/profile_round.sh: rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE|SQ_...} --output-format csv -- python3 bench.py "
               "--steps 300 --warmup 100 --no-cpu-baseline --no-dense-a (700 untimed roll-in steps; one pass per counter group, MRS_BENCH_PREWARM_S=0)",
    "correction": "FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md HBM section; calibrated in profiles/README.md); counters are KB",
    "kernel": step,
    "raw": raw,
    "fetch_bytes_per_step": fetch,
    "write_bytes_per_step": write,
    "mrs_step_bytes_per_launch": fetch + write,
    "algorithmic_bytes_per_launch": ALGO,
    "valu_busy": valu_busy,
    "valu_instructions_per_wave": sq["SQ_INSTS_VALU"] / sq["SQ_WAVES"],
    "sq": sq,
}
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
shutil.copy(os.path.join(src, "stats", "run_kernel_stats.csv"), os.path.join(ROOT, "profiles", "%s_kernel_stats.csv" % name))
shutil.copy(os.path.join(src, "bench.json"), os.path.join(ROOT, "profiles", "%s_bench.json" % name))
print("VALU busy %.3f, %.0f VALU instructions per wave" % (valu_busy, sq["SQ_INSTS_VALU"] / sq["SQ_WAVES"]))
print("fetch %.1f MB + write %.1f MB = %.1f MB per step (algorithmic %.2f MB, x%.2f)" % (fetch / 1e6, write / 1e6, (fetch + write) / 1e6, ALGO / 1e6, (fetch + write) / ALGO))
