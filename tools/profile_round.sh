#!/bin/bash
# Round profile on the GPU box: kernel stats + the two HBM counter passes, each its own rocprofv3 run.
# usage: tools/profile_round.sh <tag>   -> gpurun_out/prof_<tag>/{stats,FETCH_SIZE,WRITE_SIZE}
set -e
tag=$1
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
export MRS_BENCH_PREWARM_S=0   # profile the measured swarm only
rocprofv3 --kernel-trace --stats --output-format csv -d "$root/gpurun_out/prof_$tag/stats" -o run -- python3 "$root/bench.py" --steps 1000 --warmup 100 --no-cpu-baseline > "$root/gpurun_out/prof_$tag/bench.json" 2> "$root/gpurun_out/prof_$tag/stats.err"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$root/gpurun_out/prof_$tag/$c" -o run -- python3 "$root/bench.py" --steps 300 --warmup 700 --no-cpu-baseline > /dev/null 2> "$root/gpurun_out/prof_$tag/$c.err"
done
