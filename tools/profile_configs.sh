#!/bin/bash
# GPU box: rocprofv3 evidence for the BASELINE configurations that are not the bench line (tools/steady_bench.py workload:
# 700 roll-in steps, then 300 launches): kernel stats, HBM traffic (FETCH_SIZE / WRITE_SIZE in their own passes) and the SQ
# counters, per configuration.  usage: tools/profile_configs.sh   -> gpurun_out/prof_cfg/<tag>.txt
set -e
root=$(pwd)
out="$root/gpurun_out/prof_cfg"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
export K=100 REPS=3
run() {
  tag=$1; shift
  export "$@"
  f="$out/$tag.txt"
  echo "# $tag: $* (tools/steady_bench.py, 700 roll-in + 300 launches)" > "$f"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/$tag.stats" -o run -- python3 "$root/tools/steady_bench.py" $tag >> "$f" 2>/dev/null
  echo "## rocprofv3 --kernel-trace --stats (kernel rows)" >> "$f"
  grep -E "^\"Name\"|k_step|k_contact|k_observe" "$out/$tag.stats/run_kernel_stats.csv" >> "$f"
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$out/$tag.$c" -o run -- python3 "$root/tools/steady_bench.py" $tag > /dev/null 2>&1
    echo "## --pmc $c (KB per launch, mean of the last 100 launches; FETCH_SIZE reads half the bytes on gfx950, see profiles/README.md)" >> "$f"
    python3 "$root/tools/pmc_summary.py" "$out/$tag.$c/run_counter_collection.csv" 100 | grep -A2 -E "k_step|k_contact|k_observe" >> "$f"
  done
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d "$out/$tag.SQ" -o run -- python3 "$root/tools/steady_bench.py" $tag > /dev/null 2>&1
  echo "## --pmc SQ_* (per launch, mean of the last 100)" >> "$f"
  python3 "$root/tools/pmc_summary.py" "$out/$tag.SQ/run_counter_collection.csv" 100 | grep -A9 -E "k_step|k_contact|k_observe" >> "$f"
  unset NOADJ
  echo "$tag done"
}
run C2 E=1024 N=64 ATYPE=set_speeds NOADJ=1
run C4 E=1024 N=256 ATYPE=set_control
run C5share E=4096 N=64 ATYPE=set_target_pos
run C3 E=4096 N=64 ATYPE=set_target_vel
