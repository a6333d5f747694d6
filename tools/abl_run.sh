#!/bin/bash
# GPU box: steady-state timing of ablation builds, two interleaved rounds.  usage: tools/abl_run.sh <outfile> name1 name2 ...
# (a name may carry environment settings: "name:VAR=value")
out=$1; shift
mkdir -p "$(dirname $out)"
for round in 1 2; do
  for spec in "$@"; do
    n=${spec%%:*}; envs=""; [ "$spec" != "$n" ] && envs=${spec#*:}
    env $envs MRS_HIP_LIB=build/abl/lib$n.so timeout -k 5 200 python tools/steady_bench.py $spec 2>&1 | grep -v amdgpu.ids >> $out || echo "$spec FAILED" >> $out
  done
done
cat $out
