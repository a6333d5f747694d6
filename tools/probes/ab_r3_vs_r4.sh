#!/bin/bash
# round 4: is the step kernel itself slower than round 3's at equal sweep caps?  (build/abl/libr3.so = the sources of 17b9619 with the
# ABI number patched; libhead.so = HEAD; nodiv / restonly / both = HEAD without the division guard / the flat-body closed forms;
# oldprio = round 3's priority ladder.)  Same box, three interleaved rounds, everything at the sweep cap 10.
export K=400 REPS=3 SOLVER_ITERS=10
mkdir -p gpurun_out/r4y
tools/abl_run3.sh gpurun_out/r4y/c3b.txt r3 head nodiv restonly both oldprio both_oldprio
