#!/bin/bash
# Instruction mix + stall counters of the step kernel at steady state (GPU box; one rocprofv3 --pmc run per counter group,
# kernel trace only).  usage: tools/probes/pmc_mix.sh <tag> [lib.so]   -> gpurun_out/pmc_<tag>/<group>/
set -e
tag=$1; lib=${2:-}
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
[ -n "$lib" ] && export MRS_HIP_LIB="$root/$lib"
export ROLLIN=700 K=100 REPS=1
g1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT"
g2="SQ_WAVES SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64"
g3="SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY"
i=0; mkdir -p "$root/gpurun_out/pmc_$tag"
for g in "$g1" "$g2" "$g3"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $g --output-format csv -d "$root/gpurun_out/pmc_$tag/g$i" -o run -- python3 "$root/tools/steady_bench.py" pmc_$tag > "$root/gpurun_out/pmc_$tag/g$i.out" 2> "$root/gpurun_out/pmc_$tag/g$i.err" || { echo "pass $i failed"; tail -5 "$root/gpurun_out/pmc_$tag/g$i.err"; }
  echo "pass $i done"
done
