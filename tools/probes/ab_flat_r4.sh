#!/bin/bash
# round 4: the flat-body closed forms at the flatness bound 1e-6 (head) and 3e-8 (e38) against round 3's rest-only shortcut (restonly)
export K=300 REPS=3
mkdir -p gpurun_out/r4w
tools/abl_run3.sh gpurun_out/r4w/c3.txt head e38 restonly
E=1024 N=256 ATYPE=set_control tools/abl_run.sh gpurun_out/r4w/c4.txt head e38 restonly
E=1024 N=64 ATYPE=set_speeds NOADJ=1 tools/abl_run.sh gpurun_out/r4w/c2.txt head e38 restonly
E=4096 N=64 ATYPE=set_target_pos tools/abl_run.sh gpurun_out/r4w/c5.txt head e38 restonly
