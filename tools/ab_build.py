#!/usr/bin/env python3
"""Builds library variants for same-box A/B timing on the GPU box:
     tools/ab_build.py base:HEAD~1 new            ->  build/variants/base.so (the sources of that git revision), build/variants/new.so (working tree)
(run locally; build/ travels with the gpurun snapshot; tools/ab_run.sh runs bench.py once per variant through PWN_HIP_LIB).  The product sources
carry no compile-time switches (tests/test_capi_cpu.py): a variant is another revision of the sources, built with the product flags."""
import os
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from g2o_frontend_amd import build as b  # noqa: E402


def one(spec):
    name, _, rev = spec.partition(":")
    out = os.path.join(ROOT, "build", "variants", name + ".so")
    if not rev:
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + b.FLAGS + ["-o", out, b.SOURCES[0]])
        return out
    with tempfile.TemporaryDirectory(prefix="pwn_ab_") as d:
        tar = subprocess.run(["git", "-C", ROOT, "archive", rev, "g2o_frontend_amd/csrc", "include"], check=True, capture_output=True).stdout
        subprocess.run(["tar", "-x", "-C", d], input=tar, check=True)
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + b.FLAGS + ["-o", out, os.path.join(d, "g2o_frontend_amd", "csrc", "pwn_hip_capi.hip")])
    return out


if __name__ == "__main__":
    d = os.path.join(ROOT, "build", "variants")
    os.makedirs(d, exist_ok=True)
    for f in os.listdir(d):
        os.remove(os.path.join(d, f))
    with ThreadPoolExecutor(4) as ex:
        for o in ex.map(one, sys.argv[1:]):
            print(o)
