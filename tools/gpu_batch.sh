#!/bin/bash
# Run several GPU steps in one gpurun call: `tools/gpu_batch.sh OUTDIR 'name|timeout_s|command' ...`.
# A step that merely FAILS (exit 1: a failing test) does not stop the batch; a step that times out or is killed does
# (no further GPU step is started after a hang).  Every step's output goes to OUTDIR/name.log.
out=$1; shift
mkdir -p "$out"
for spec in "$@"; do
  name=${spec%%|*}; rest=${spec#*|}; tmo=${rest%%|*}; cmd=${rest#*|}
  echo "=== $name (timeout ${tmo}s): $cmd"
  t0=$(date +%s)
  timeout -k 10 "$tmo" bash -c "$cmd" > "$out/$name.log" 2>&1
  rc=$?
  echo "=== $name rc=$rc after $(( $(date +%s) - t0 )) s"; tail -n 6 "$out/$name.log"
  if [ $rc -ge 124 ]; then echo "=== stopping the batch: $name was killed or timed out"; exit $rc; fi
done
exit 0
