#!/bin/bash
# round 4: flatness bound of the closed-form contact shortcut (MRS_REST_EPS): per-step error against the oracle (teacher-forced,
# C3 and C2, 400 steps) and the step's duration, per build
mkdir -p gpurun_out/r4t
for n in e6 e37 e7 e38 restonly; do
  echo "=== $n" >> gpurun_out/r4t/probe.txt
  MRS_HIP_LIB=build/abl/lib$n.so timeout -k 5 200 python tools/teacher_probe.py --configs C3,C2,C4 --steps 500 2>&1 | grep -v "amdgpu\|\.\.\." >> gpurun_out/r4t/probe.txt
done
export K=300 REPS=3
tools/abl_run.sh gpurun_out/r4t/c3.txt e6 e37 e7 e38 restonly
E=1024 N=256 ATYPE=set_control tools/abl_run.sh gpurun_out/r4t/c4.txt e6 e37 e7 e38 restonly
E=1024 N=64 ATYPE=set_speeds NOADJ=1 tools/abl_run.sh gpurun_out/r4t/c2.txt e6 e37 e7 e38 restonly
