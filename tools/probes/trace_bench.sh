#!/bin/bash
# Diagnostic (GPU box): kernel trace of the default bench protocol; prints k_step durations and gaps per 50 launches.
set -e
root=$(pwd); out="$root/gpurun_out/trace_bench"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
export MRS_BENCH_PREWARM_S=${PREWARM:-0}
rocprofv3 --kernel-trace --output-format csv -d "$out/t" -o run -- python3 "$root/bench.py" --no-cpu-baseline --no-dense-a > "$out/bench.json" 2> "$out/err.txt"
python3 - "$out/t/run_kernel_trace.csv" <<'P'
import csv, sys, numpy as np
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_step' in r['Kernel_Name']]
s = np.array([int(rows[i]['Start_Timestamp']) for i in idx]); e = np.array([int(rows[i]['End_Timestamp']) for i in idx])
d = (e - s) / 1e3; gap = np.r_[0, (s[1:] - e[:-1]) / 1e3]
print(len(idx), "k_step launches")
for a in range(0, len(idx), 50):
    others = set()
    for n in range(a, min(a + 50, len(idx))):
        lo = idx[n - 1] + 1 if n else 0
        others |= {rows[j]['Kernel_Name'][:30] for j in range(lo, idx[n])}
    print("%5d  dur %.2f [%.2f..%.2f]  gap sum %.1f max %.1f  %s" % (a, d[a:a+50].mean(), d[a:a+50].min(), d[a:a+50].max(), gap[a:a+50].sum(), gap[a:a+50].max(), sorted(others)[:3]))
P
