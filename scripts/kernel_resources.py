#!/usr/bin/env python
"""Prints VGPRs / scratch / occupancy of every kernel of a .hip source (hipcc -Rpass-analysis=kernel-resource-usage).
    python scripts/kernel_resources.py gnngls_amd/csrc/gls_kernels.hip [filter] [-- extra hipcc flags]"""
import re, subprocess, sys, tempfile, os
src = sys.argv[1]
rest = sys.argv[2:]
extra = []
if "--" in rest:
    k = rest.index("--"); extra = rest[k + 1:]; rest = rest[:k]
flt = rest[0] if rest else ""
with tempfile.TemporaryDirectory() as td:
    out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                          "-c", src, "-o", os.path.join(td, "x.o"), "-Rpass-analysis=kernel-resource-usage"] + extra,
                         capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+(SGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).split()[0]] = int(m.group(2))
for name, r in rows.items():
    if flt in name:
        short = re.sub(r"\(.*", "", name).replace("gnngls::", "")
        print(f"{short:70s} vgpr {r.get('VGPRs')} agpr {r.get('AGPRs', 0)} sgpr {r.get('SGPRs')} scratch {r.get('ScratchSize')} occ {r.get('Occupancy')}")
