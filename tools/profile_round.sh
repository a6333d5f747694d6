#!/bin/bash
# Round profile on the GPU box: kernel stats + the HBM counter passes + the SQ pass, each its own rocprofv3 run
# (counters never together with --stats; the program itself after "--").
# usage: tools/profile_round.sh <tag>   -> gpurun_out/prof_<tag>/{stats,FETCH_SIZE,WRITE_SIZE,SQ}
set -e
tag=$1
root=$(pwd)
out="$root/gpurun_out/prof_$tag"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
# the device warm-up runs on another instantiation of the step kernel (k_step<set_target_pos>): the rows of the measured
# kernel hold the measured swarm's 700 roll-in + warm-up + timed launches only, at the clock a sustained load settles at
# (without any warm-up the 1800 launches of a pass are a profile of the clock ramp after an idle start: 26 -> 21 us)
export MRS_BENCH_PREWARM_S=0.5 MRS_BENCH_PREWARM_ATYPE=set_target_pos
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o run -- python3 "$root/bench.py" --steps 1000 --warmup 100 --no-cpu-baseline --no-dense-a --no-model-legs > "$out/bench.json" 2> "$out/stats.err"
echo "stats done"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$out/$c" -o run -- python3 "$root/bench.py" --steps 300 --warmup 100 --no-cpu-baseline --no-dense-a --no-model-legs > /dev/null 2> "$out/$c.err"
  echo "$c done"
done
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d "$out/SQ" -o run -- python3 "$root/bench.py" --steps 300 --warmup 100 --no-cpu-baseline --no-dense-a --no-model-legs > /dev/null 2> "$out/SQ.err"
echo "SQ done"
