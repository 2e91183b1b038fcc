#!/usr/bin/env python3
"""Registers / scratch / occupancy of every kernel of one translation unit, from hipcc's resource-usage remarks.
usage: python tools/kernel_resources.py csrc/nfp_tile.hip [name-filter]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neighbour_feature_pooling_amd import build
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run([build.hipcc_path()] + build.HIPCC_FLAGS + ["-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/dev/null", src],
                     capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark: (?:\s*)Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0]
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs|LDS Size \[bytes/block\]): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).split(" ")[0]] = int(m.group(2))
    if "error" in line:
        print(line)
for k, v in sorted(rows.items()):
    if flt in k:
        print(f"{k:70s} VGPR {v.get('VGPRs', -1):4d}  scratch {v.get('ScratchSize', -1):4d}  waves/SIMD {v.get('Occupancy', -1)}  SGPR {v.get('SGPRs', -1)}")
