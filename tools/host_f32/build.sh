#!/bin/bash
# Diagnostic build (CPU): the kernel's float32 contact solve for the host.  tools/host_f32/build.sh -> build/libcontact_host.so
# shim/hip/hip_runtime.h stands in for the HIP header and maps the gfx950 builtins the device functions use; the contact rows are a
# template on their scalar type (mrs_device.hpp contact_solve_rows<S>): float as in the kernels, double for the float64 comparison.
# The library is written under a temporary name and moved into place (concurrent pytest workers).
set -e
cd "$(dirname "$0")"
mkdir -p ../../build
tmp=../../build/libcontact_host.so.$$
/opt/rocm/lib/llvm/bin/clang++ -O2 -std=c++17 -fPIC -shared -ffp-contract=off -DMRS_HOST_CHECK -Ishim contact_host.cpp -o $tmp
mv -f $tmp ../../build/libcontact_host.so
echo built build/libcontact_host.so
