#!/bin/bash
# GPU box: steady-state timing of ablation builds, two interleaved rounds.  usage: tools/abl_run.sh <outfile> name1 name2 ...
out=$1; shift
mkdir -p "$(dirname $out)"
for round in 1 2; do
  for n in "$@"; do
    MRS_HIP_LIB=build/abl/lib$n.so timeout -k 5 200 python tools/steady_bench.py $n 2>&1 | grep -v amdgpu.ids >> $out || echo "$n FAILED" >> $out
  done
done
cat $out
