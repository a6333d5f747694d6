#!/bin/bash
mkdir -p gpurun_out/r3t
F="--no-cpu-baseline --no-dense-a --no-model-legs"
for cfg in "10 50" "50 50" "25 50" "10 500" "40 50" "10 50"; do
  set -- $cfg
  MRS_BENCH_EVENT_SPAN=$1 MRS_BENCH_EVENT_EVERY=$2 python bench.py $F 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('span $1 every $2: ms_per_step %.3f us  kernel %.3f us  frac %.4f' % (d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3, d['roofline']['frac']))"
done
