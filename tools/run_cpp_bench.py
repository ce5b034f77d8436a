#!/usr/bin/env python3
"""Writes a few seeded synthetic VGA depth pairs as 16-bit PGM files and runs tools/pwn_hip_bench (bench.py's step driven from C++) on them.
usage: python tools/run_cpp_bench.py [pairs=128] [steps=5] [distinct=8]"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from g2o_frontend_amd import build, synth
    build.build(); build.build_tools()
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    distinct = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    d = tempfile.mkdtemp(prefix="pwn_bench_")
    names = []
    for i in range(distinct):
        r, c, _ = synth.make_pair(i, 480, 640, synth.K_VGA)
        for tag, img in (("r", r), ("c", c)):
            fn = os.path.join(d, f"{tag}{i}.pgm")
            with open(fn, "wb") as f:
                f.write(b"P5\n%d %d\n65535\n" % (img.shape[1], img.shape[0])); f.write(img.astype(">u2").tobytes())
            names.append(fn)
    lst = os.path.join(d, "list.txt")
    open(lst, "w").write("\n".join(names) + "\n")
    for mode in (0, 1, 2):      # resident frames / host frames in one page-locked block / double-buffered upload by the caller
        out = subprocess.check_output([os.path.join(ROOT, "tools", "pwn_hip_bench"), lst, str(P), str(steps), "2", "0", str(mode)]).decode()
        print(out.splitlines()[0])


if __name__ == "__main__":
    main()
