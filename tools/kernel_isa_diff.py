#!/usr/bin/env python3
"""Per-kernel ISA comparison of two builds of libpwn_hip.so (gfx950 code object, llvm-objdump -d, addresses and branch targets dropped):
  tools/kernel_isa_diff.py old.so new.so
prints, per kernel of the old build, whether the new build holds the same instruction sequence.  Used when the kernel header is edited without
an intended change of the generated code (e.g. removing compile-time switches): identical ISA = identical results."""
import re
import struct
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def kernels(path):
    blob = open(path, "rb").read()
    i = blob.find(b"__CLANG_OFFLOAD_BUNDLE__")
    n = struct.unpack_from("<Q", blob, i + 24)[0]; off = i + 32
    code = None
    for _ in range(n):
        o, sz, tl = struct.unpack_from("<QQQ", blob, off); name = blob[off + 24: off + 24 + tl].decode(); off += 24 + tl
        if "gfx950" in name:
            code = blob[i + o: i + o + sz]
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(code); f.flush()
        asm = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True).stdout
    out, cur = {}, None
    for line in asm.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur = m.group(1); out[cur] = []; continue
        if cur is None or not line.strip():
            continue
        ins = re.sub(r"//.*$", "", line).strip()
        ins = re.sub(r"^\s*[0-9a-f]+:\s*", "", ins)
        if ins:
            out[cur].append(ins)
    return out


if __name__ == "__main__":
    a, b = kernels(sys.argv[1]), kernels(sys.argv[2])
    same = diff = 0
    for k in sorted(a):
        if k not in b:
            print("GONE   ", k, len(a[k])); continue
        if a[k] == b[k]:
            same += 1; print("same   ", k, len(a[k]))
        else:
            diff += 1
            nd = sum(1 for x, y in zip(a[k], b[k]) if x != y) + abs(len(a[k]) - len(b[k]))
            print("DIFFERS", k, len(a[k]), len(b[k]), "differing lines ~", nd)
    for k in sorted(b):
        if k not in a:
            print("NEW    ", k, len(b[k]))
    print(f"{same} kernels identical, {diff} differ")
