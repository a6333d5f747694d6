#!/usr/bin/env python3
"""Registers / scratch / occupancy of every kernel of the library as the compiler reports them (no GPU needed):
    python tools/kernel_resources.py [extra hipcc flags ...] | grep k_step"""
import os, re, subprocess, sys, shutil, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = tempfile.mktemp(suffix=".so")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-slp-vectorize", "-mllvm", "-amdgpu-kernarg-preload-count=14"] + sys.argv[1:] + [
    "-Rpass-analysis=kernel-resource-usage", os.path.join(ROOT, "mrs-gym_amd/csrc/mrs_kernels.hip"), "-o", out]
txt = subprocess.run(cmd, capture_output=True, text=True).stderr
if os.path.exists(out):
    os.remove(out)
filt = shutil.which("c++filt") or shutil.which("llvm-cxxfilt")
name, row = None, {}
for line in txt.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = m.group(1)
        if filt:
            name = subprocess.run([filt, name], capture_output=True, text=True).stdout.strip() or name
        row = {}
        continue
    m = re.search(r"remark:\s+(VGPRs|TotalSGPRs|ScratchSize|Occupancy|LDS Size)(?: \[[^\]]*\])?: (\d+)", line)
    if m and name:
        row[m.group(1)] = int(m.group(2))
        if m.group(1) == "LDS Size":
            print("%-78s VGPR %3d SGPR %3d scratch %4d occupancy %d" % (name[:78], row.get("VGPRs", -1), row.get("TotalSGPRs", -1), row.get("ScratchSize", -1), row.get("Occupancy", -1)))
