#!/bin/bash
# as tools/abl_run.sh, three interleaved rounds
out=$1; shift
mkdir -p "$(dirname $out)"; rm -f $out
for round in 1 2 3; do
  for spec in "$@"; do
    n=${spec%%:*}; envs=""; [ "$spec" != "$n" ] && envs=${spec#*:}
    env $envs MRS_HIP_LIB=build/abl/lib$n.so timeout -k 5 200 python tools/steady_bench.py $spec 2>&1 | grep -v amdgpu.ids >> $out || echo "$spec FAILED" >> $out
  done
done
sort -s -k1,1 $out
