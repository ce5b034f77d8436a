#!/usr/bin/env python3
"""Instruction histogram of one kernel of libpwn_hip.so (gfx950 code object): python tools/kernel_isa_hist.py k_stats [lib.so]"""
import collections, os, re, struct, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "g2o_frontend_amd", "libpwn_hip.so")
blob = open(lib, "rb").read()
i = blob.find(b"__CLANG_OFFLOAD_BUNDLE__")
n = struct.unpack_from("<Q", blob, i + 24)[0]; off = i + 32; code = None
for _ in range(n):
    o, sz, tl = struct.unpack_from("<QQQ", blob, off); name = blob[off + 24: off + 24 + tl].decode(); off += 24 + tl
    if "gfx950" in name:
        code = blob[i + o: i + o + sz]
with tempfile.NamedTemporaryFile(suffix=".co") as f:
    f.write(code); f.flush()
    asm = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", f.name], capture_output=True, text=True).stdout
cur = None; hist = collections.Counter(); total = 0
for line in asm.splitlines():
    m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
    if m:
        cur = m.group(1); continue
    if cur and sys.argv[1] in cur and not cur.endswith(".kd"):
        t = line.split()
        if t and re.match(r"^[sv]_|^global_|^ds_|^buffer_|^flat_", t[0]):
            hist[t[0]] += 1; total += 1
print(total, "instructions")
groups = collections.Counter()
for k, v in hist.items():
    g = "f64" if "f64" in k else ("div/rcp/sqrt f32" if re.search(r"div_|rcp|sqrt|rsq", k) else ("valu" if k.startswith("v_") else ("salu" if k.startswith("s_") else "mem")))
    groups[g] += v
print(dict(groups))
for k, v in hist.most_common(45):
    print(f"{v:5d} {k}")
