#!/bin/bash
# One rocprofv3 counter pass over a short bench run (GPU box; counters in their own run, kernel trace only).
# usage: tools/probes/pmc_pass.sh <tag> <counter> [<counter> ...]   -> gpurun_out/pmc_<tag>/
set -e
tag=$1; shift
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
MRS_BENCH_PREWARM_S=0 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$root/gpurun_out/pmc_$tag" -o run -- python3 "$root/bench.py" --no-cpu-baseline --steps ${PMC_STEPS:-300} --warmup ${PMC_WARMUP:-100} > "$root/gpurun_out/pmc_$tag.json" 2> "$root/gpurun_out/pmc_$tag.err"
