#!/bin/bash
# Build ablation variants of the library: tools/abl_build.sh name1 "-DFLAG=1 ..." name2 "..." ...  -> build/abl/lib<name>.so
set -e
mkdir -p build/abl
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -fno-slp-vectorize ${MRS_PRELOAD--mllvm -amdgpu-kernarg-preload-count=14}"
pids=()
while [ $# -gt 0 ]; do
  name=$1; defs=$2; shift 2
  ( hipcc $FLAGS $defs mrs-gym_amd/csrc/mrs_kernels.hip -o build/abl/lib$name.so 2> build/abl/$name.log || echo "BUILD FAILED: $name" ) &
  pids+=($!)
  if [ ${#pids[@]} -ge 4 ]; then wait ${pids[0]}; pids=("${pids[@]:1}"); fi
done
wait
ls -la build/abl/*.so | awk '{print $5, $9}'
grep -l "error" build/abl/*.log 2>/dev/null || true
