#!/bin/bash
# GPU box: kernel trace of the driver's form of the bench (5 warm-up + 20 timed steps behind the 700-step roll-in):
# duration of each of the last 30 launches of the measured kernel and the gap in front of it.
root=$(pwd); out="$root/gpurun_out/short_trace"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$out" -o run -- python3 "$root/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-dense-a --no-model-legs > "$out/bench.json" 2> "$out/err.txt"
python3 - "$out/run_kernel_trace.csv" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kernel_Name"].replace("void ", "").startswith("k_step<4")]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-30:]
prev_end = int(rows[-31]["End_Timestamp"])
for k, r in enumerate(last):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("launch %4d: gap %8.2f us  duration %6.2f us" % (len(rows) - 30 + k, (s - prev_end) / 1e3, (e - s) / 1e3))
    prev_end = e
PY
