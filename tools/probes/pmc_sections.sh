#!/bin/bash
# Dynamic VALU instructions per wave of the step kernel with parts of the step switched off (GPU box; one rocprofv3
# --pmc pass per configuration).  usage: tools/probes/pmc_sections.sh <tag> <lib.so>  -> gpurun_out/pmc_<tag>/<config>/
set -e
tag=$1; lib=$2
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
export MRS_HIP_LIB="$root/$lib" ROLLIN=700 K=100 REPS=1
mkdir -p "$root/gpurun_out/pmc_$tag"
run() { # name, extra env
  name=$1; shift
  ( export "$@"; rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d "$root/gpurun_out/pmc_$tag/$name" -o run -- python3 "$root/tools/steady_bench.py" $name > "$root/gpurun_out/pmc_$tag/$name.out" 2> "$root/gpurun_out/pmc_$tag/$name.err" ) || { echo "$name failed"; tail -5 "$root/gpurun_out/pmc_$tag/$name.err"; }
  echo "$name done"
}
run full X=0
run nocontact NOCONTACT=1
run set_speeds ATYPE=set_speeds
run set_target_accel ATYPE=set_target_accel
run set_speeds_nocontact ATYPE=set_speeds NOCONTACT=1
