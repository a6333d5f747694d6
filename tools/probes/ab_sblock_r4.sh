#!/bin/bash
# round 4: workgroup size of the N = 64 step at the library's sweep cap of 10 (round 3 chose 256 at a cap of 6)
export K=400 REPS=3
rm -f gpurun_out/r4f/sblock.txt; mkdir -p gpurun_out/r4f
for round in 1 2; do
  for sb in 256 512 128; do
    MRS_STEP_BLOCK=$sb timeout -k 5 200 python tools/steady_bench.py sblock$sb 2>&1 | grep -v amdgpu.ids >> gpurun_out/r4f/sblock.txt
  done
done
cat gpurun_out/r4f/sblock.txt
