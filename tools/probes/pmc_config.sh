#!/bin/bash
# Diagnostic (GPU box): SQ counters of the step kernel for one configuration: tools/probes/pmc_config.sh <tag> <E> <N> <action type>
set -e
tag=$1; E=$2; N=$3; at=$4
root=$(pwd); out="$root/gpurun_out/pmc_$tag"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY --output-format csv -d "$out/t" -o run -- python3 "$root/tools/probes/n_block.py" $E $N $at > "$out/out.txt" 2> "$out/err.txt"
python3 "$root/tools/pmc_summary.py" "$out/t/run_counter_collection.csv" 100 | grep -A9 "k_step"
