#!/bin/bash
# round 4: the flat-body closed forms (head) against round 3's rest-only shortcut (restonly), per BASELINE config
export K=300 REPS=3
mkdir -p gpurun_out/r4r
tools/abl_run.sh gpurun_out/r4r/c3.txt head restonly
E=1024 N=64 ATYPE=set_speeds NOADJ=1 tools/abl_run.sh gpurun_out/r4r/c2.txt head restonly
E=1024 N=256 ATYPE=set_control tools/abl_run.sh gpurun_out/r4r/c4.txt head restonly
E=4096 N=64 ATYPE=set_target_pos tools/abl_run.sh gpurun_out/r4r/c5.txt head restonly
