#!/bin/bash
# the driver's form of the bench (--steps 20 --warmup 5) under runtime wait-mode settings, alternating
for i in 1 2 3 4 5 6; do for m in "X=0" "HSA_ENABLE_INTERRUPT=0"; do env $m python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dense-a 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"%-26s\" % sys.argv[1], \"%.4g\" % d[\"value\"], \"ms/step %.2f\" % (d[\"ms_per_step\"]*1e3), \"kernel %.2f\" % (d[\"roofline\"][\"kernel_ms\"]*1e3), \"host %.2f\" % (d[\"host_ms_per_step\"]*1e3))" "[$m]"; done; done
