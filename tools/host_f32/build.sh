#!/bin/bash
# Diagnostic build (CPU): the kernel's float32 contact solve for the host.  tools/host_f32/build.sh -> build/libcontact_host.so
# gen_variant.py derives two templated forms of contact_solve_f32 from mrs_device.hpp (scalar type as a parameter; set-up and
# sweeps in separate types); shim/hip/hip_runtime.h stands in for the HIP header and maps the gfx950 builtins the function uses.
set -e
cd "$(dirname "$0")"
python3 gen_variant.py
mkdir -p ../../build
/opt/rocm/lib/llvm/bin/clang++ -O2 -std=c++17 -fPIC -shared -ffp-contract=off -DMRS_HOST_CHECK -Ishim contact_host.cpp -o ../../build/libcontact_host.so
echo built build/libcontact_host.so
