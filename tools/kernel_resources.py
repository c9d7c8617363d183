#!/usr/bin/env python3
"""Register / scratch / occupancy table of every trace_kernel instantiation (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/kernel_resources.py [extra hipcc flags]   (cross-compiles for gfx950, no GPU needed)"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "raytracingweekend.jl_amd", "csrc", "rtw_launch.hip")
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "-mllvm", "-amdgpu-mfma-vgpr-form",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + sys.argv[1:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}; rows.append(cur); continue
    m = re.search(r"remark:\s+(TotalSGPRs|VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill): (\d+)", line)
    if m and cur is not None:
        cur[m.group(1)] = int(m.group(2))
print(f"{'kernel':52s} {'VGPR':>5s} {'SGPR':>5s} {'waves':>5s} {'scratch B':>9s} {'vspill':>6s} {'sspill':>6s}")
for r in rows:
    n = r["name"]
    m = re.match(r"_ZN3rtw12trace_kernelI([fd])Lb([01])ELb([01])ELb([01])ELb([01])ELi(n?\d+)E", n)
    if m:
        label = f"trace<{'f32' if m.group(1) == 'f' else 'f64'}{', profile' if m.group(2) == '1' else ''}{', lds-scene' if m.group(3) == '1' else ', global-scene'}{', cull' if m.group(4) == '1' else ''}{', mfma' if m.group(5) == '1' else ''}{', numerics fixed' if not m.group(6).startswith('n') else ''}>"
    elif "unit_kernel" in n:
        label = "unit_kernel<%s>" % ("f32" if "IfE" in n else "f64")
    else:
        label = n[:44]
    print(f"{label:52s} {r.get('VGPRs', 0):5d} {r.get('TotalSGPRs', 0):5d} {r.get('Occupancy [waves/SIMD]', 0):5d} {r.get('ScratchSize [bytes/lane]', 0):9d} {r.get('VGPRs Spill', 0):6d} {r.get('SGPRs Spill', 0):6d}")
