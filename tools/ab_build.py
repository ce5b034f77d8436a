#!/usr/bin/env python3
"""Builds library variants for A/B timing on the GPU box: tools/ab_build.py name1:-DFLAG=1,-DX=2 name2: ...  ->  build/variants/<name>.so
(run locally; build/ travels with the gpurun snapshot).  `base` = the product flags."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from g2o_frontend_amd import build as b  # noqa: E402


def one(spec):
    name, _, flags = spec.partition(":")
    out = os.path.join(ROOT, "build", "variants", name + ".so")
    extra = [f for f in flags.split(",") if f]
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + b.FLAGS + extra + ["-o", out, b.SOURCES[0]])
    return out


if __name__ == "__main__":
    d = os.path.join(ROOT, "build", "variants")
    os.makedirs(d, exist_ok=True)
    for f in os.listdir(d):
        os.remove(os.path.join(d, f))
    with ThreadPoolExecutor(4) as ex:
        for o in ex.map(one, sys.argv[1:]):
            print(o)
