#!/bin/bash
# One rocprofv3 counter pass over tools/steady_bench.py (GPU box; counters in their own run, kernel trace only).
# usage: tools/probes/pmc_steady.sh <tag> <counter> [<counter> ...]   -> gpurun_out/pmc_<tag>/, summary on stdout
set -e
tag=$1; shift
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
K=${K:-100} REPS=1 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$root/gpurun_out/pmc_$tag" -o run -- python3 "$root/tools/steady_bench.py" $tag > "$root/gpurun_out/pmc_$tag.txt" 2> "$root/gpurun_out/pmc_$tag.err"
python3 "$root/tools/pmc_summary.py" $(ls "$root"/gpurun_out/pmc_$tag/*counter_collection.csv | head -1) 100
