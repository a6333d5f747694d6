"""Summarise a rocprofv3 --kernel-trace CSV of bench.py by phase of the run: mean / min / max duration of the step kernel
over the roll-in launches, the warm-up launches and the timed launches.

    python tools/trace_phases.py gpurun_out/prof_<tag>/stats/run_kernel_trace.csv [rollin=700] [warmup=100]

(The --stats average mixes the three.)"""
import csv, sys
import numpy as np
path = sys.argv[1]
rollin = int(sys.argv[2]) if len(sys.argv) > 2 else 700
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 100
# k_step<4, ...> = set_target_vel, the measured swarm (the device warm-up of tools/profile_round.sh runs k_step<5, ...>)
rows = [r for r in csv.DictReader(open(path)) if r["Kernel_Name"].replace("void ", "").startswith("k_step<4")]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = np.array([(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows])
gap = np.array([(int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 for a, b in zip(rows[:-1], rows[1:])])
name = rows[0]["Kernel_Name"]
print("%s: %d launches, mean %.2f us" % (name, len(d), d.mean()))
for label, lo, hi in (("roll-in", 0, rollin), ("warm-up", rollin, rollin + warm), ("timed", rollin + warm, len(d))):
    x = d[lo:hi]
    if len(x):
        g = gap[lo:hi - 1] if hi - 1 > lo else np.array([0.0])
        print("  %-8s launches %5d..%5d: mean %6.2f us  median %6.2f  min %6.2f  max %6.2f | gap to the next launch: median %.2f us" % (
            label, lo, hi - 1, x.mean(), np.median(x), x.min(), x.max(), np.median(g)))
algo = 268 * 4096 * 64
x = d[rollin + warm:]
if len(x):
    print("  timed launches: %.1f MB algorithmic / %.2f us = %.0f GB/s = %.3f of 8 TB/s" % (algo / 1e6, x.mean(), algo / x.mean() / 1e3, algo / x.mean() / 1e3 / 8000))
