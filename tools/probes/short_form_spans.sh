#!/bin/bash
# GPU box: the driver's form of the bench (20 timed steps behind 5 warm-up steps) with the timed region cut into spans of 5
# launches: where inside the 20 steps the time goes (MRS_BENCH_DEBUG_SPANS prints the spans in order).
F="--no-cpu-baseline --no-dense-a --no-model-legs --steps 20 --warmup 5"
for i in 1 2 3; do
  MRS_BENCH_EVENT_SPAN=5 MRS_BENCH_EVENT_EVERY=5 MRS_BENCH_DEBUG_SPANS=1 python bench.py $F 2>&1 >/dev/null | grep spans
  python bench.py $F 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  default: ms_per_step %.3f us  kernel %.3f us  host %.3f us' % (d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3, d['host_ms_per_step']*1e3))"
done
F2="--no-cpu-baseline --no-dense-a --no-model-legs --steps 200 --warmup 50"
MRS_BENCH_EVENT_SPAN=10 MRS_BENCH_EVENT_EVERY=10 MRS_BENCH_DEBUG_SPANS=1 python bench.py $F2 2>&1 >/dev/null | grep spans
