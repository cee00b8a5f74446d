#!/usr/bin/env python
"""Per-kernel register / LDS / occupancy summary of a .hip file (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/kernel_resources.py gaustudio_amd/csrc/gsr_kernels_bwd.hip [name substring]"""
import os, re, subprocess, sys

f = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
       "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-gpu-flush-denormals-to-zero", "-munsafe-fp-atomics",
       "-fno-slp-vectorize",   # as in csrc/Makefile for the kernel files
       "-I" + os.path.dirname(os.path.abspath(f)), "-Rpass-analysis=kernel-resource-usage", "-c", f, "-o", "/dev/null"]
if "kernels_bwd" in f or "kernels_fwd" in f:
    cmd[1:1] = ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for l in out.splitlines():
    m = re.search(r"remark: +(.*?) \[-Rpass", l)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
for r in rows:
    n = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip().split("(")[0]
    if pat and pat not in n:
        continue
    g = lambda k: r.get(k, "?")
    print(f"{n[:72]:72s} VGPR {g('VGPRs'):>4} AGPR {g('AGPRs'):>3} SGPR {g('TotalSGPRs'):>4} scratch {g('ScratchSize [bytes/lane]'):>4} "
          f"occ {g('Occupancy [waves/SIMD]'):>2} LDS {g('LDS Size [bytes/block]')}")
