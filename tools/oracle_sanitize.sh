#!/bin/bash
# CPU only: the oracle's tests with the C restatement built under AddressSanitizer + UndefinedBehaviorSanitizer (gcc).
# (GPU sanitizers are not available on this pool; the HIP library's host code needs a GPU to run at all.)
set -e
cd "$(dirname "$0")/.."
cp oracle/libmrs_oracle.so /tmp/libmrs_oracle.keep.so 2>/dev/null || true
trap 'cp /tmp/libmrs_oracle.keep.so oracle/libmrs_oracle.so 2>/dev/null; touch oracle/libmrs_oracle.so' EXIT
( cd oracle && gcc -O1 -g -std=gnu99 -fPIC -ffp-contract=off -fno-fast-math -mfma -fopenmp -fsanitize=address,undefined -fno-omit-frame-pointer \
      -shared mrs_oracle.c mrs_sensors.c -o libmrs_oracle.so -lm )
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1 \
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) \
    python -m pytest tests/test_oracle_golden.py tests/test_oracle_physics.py tests/test_oracle_sensors.py -x -q 2>&1 | tee /tmp/oracle_sanitize.log | tail -3
echo "sanitizer reports: $(grep -c 'runtime error\|AddressSanitizer' /tmp/oracle_sanitize.log)"
