#!/usr/bin/env python3
"""Prints VGPR / scratch / LDS / occupancy per kernel of libpwn_hip (hipcc -Rpass-analysis=kernel-resource-usage)."""
import re
import subprocess
import sys
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from g2o_frontend_amd import build as b  # noqa: E402

cmd = ["/opt/rocm/bin/hipcc"] + b.FLAGS + ["-Rpass-analysis=kernel-resource-usage", "-o", "/tmp/_pwn_res.so", b.SOURCES[0]]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur, d = None, {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur, d = m.group(1), {}
        continue
    for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                     ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)"), ("sgpr", r"TotalSGPRs: (\d+)")):
        m = re.search(pat, line)
        if m:
            d[key] = int(m.group(1))
    if "LDS Size" in line and cur:
        name = re.sub(r"^_ZN6pwnhip\d+", "", cur)[:28]
        print(f"{name:30s} vgpr {d.get('vgpr'):4d} sgpr {d.get('sgpr'):4d} scratch {d.get('scratch'):5d} lds {d.get('lds'):6d} occ {d.get('occ')}")
