"""Builds the in-tree HIP extension ``libpwn_hip.so`` for gfx950 (hipcc cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = [os.path.join(_HERE, "csrc", f) for f in ("pwn_hip_capi.hip", "pwn_kernels.h", "pwn_math.h", "pwn_scene_kernels.h", "pwn_scene_capi.h", "pwn_stats.h")]
HEADER = os.path.join(os.path.dirname(_HERE), "include", "pwn_hip.h")
TEST_HEADER = os.path.join(os.path.dirname(_HERE), "include", "pwn_hip_testing.h")
OUT = os.path.join(_HERE, "libpwn_hip.so")
# -ffp-contract=off: the kernels reproduce the CPU path's evaluation order; a fused multiply-add would change bits.
# -fno-slp-vectorize: the SLP vectoriser pairs scalar fp32 ops into v_pk_*_f32 and pays for it in v_mov shuffles and registers
#   (k_corr_linearize 124 -> 98 VGPRs without it; same bits, +3 % whole-step throughput measured on MI355X).
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-shared", "-pthread", "-Wall",
         "-Wno-unused-function"]


def build(force: bool = False, verbose: bool = False) -> str:
    deps = SOURCES + [HEADER, TEST_HEADER, __file__]
    stale = (not os.path.exists(OUT)) or any(os.path.getmtime(d) > os.path.getmtime(OUT) for d in deps)
    if force or stale:
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        cmd = [hipcc] + FLAGS + ["-o", OUT, SOURCES[0]]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return OUT


def build_tools(force: bool = False) -> str:
    """The C++ harness over the host mirror (tools/pwn_hip_simple_aligner.cpp); plain g++, links libpwn_hip.so."""
    root = os.path.dirname(_HERE)
    src = os.path.join(root, "tools", "pwn_hip_simple_aligner.cpp")
    out = os.path.join(root, "tools", "pwn_hip_simple_aligner")
    deps = [src, os.path.join(_HERE, "host", "pwn_hip.hpp"), HEADER, OUT]
    if force or (not os.path.exists(out)) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", root, src, "-o", out, "-L", _HERE, "-lpwn_hip",
                               "-Wl,-rpath,$ORIGIN/../g2o_frontend_amd"])
    # the mapping loop of pwn_aligner.cpp (Cloud::add + Merger::merge + Cloud::save) over the same mirror
    src2 = os.path.join(root, "tools", "pwn_hip_scene_aligner.cpp")
    out2 = os.path.join(root, "tools", "pwn_hip_scene_aligner")
    deps2 = [src2] + deps[1:]
    if force or (not os.path.exists(out2)) or any(os.path.getmtime(d) > os.path.getmtime(out2) for d in deps2):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", root, src2, "-o", out2, "-L", _HERE, "-lpwn_hip",
                               "-Wl,-rpath,$ORIGIN/../g2o_frontend_amd"])
    # the tracker + closer flow (PwnTracker::processFrame, CloudCache, batched matchClouds, statistics, priors) over the same mirror
    src3 = os.path.join(root, "tools", "pwn_hip_tracker_app.cpp")
    out3 = os.path.join(root, "tools", "pwn_hip_tracker_app")
    deps3 = [src3] + deps[1:]
    if force or (not os.path.exists(out3)) or any(os.path.getmtime(d) > os.path.getmtime(out3) for d in deps3):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", root, src3, "-o", out3, "-L", _HERE, "-lpwn_hip",
                               "-Wl,-rpath,$ORIGIN/../g2o_frontend_amd"])
    # bench.py's workload driven from C++ (device and page-locked memory through the C-ABI: no HIP headers, no HIP runtime on the link line)
    src4 = os.path.join(root, "tools", "pwn_hip_bench.cpp")
    out4 = os.path.join(root, "tools", "pwn_hip_bench")
    deps4 = [src4] + deps[1:]
    if force or (not os.path.exists(out4)) or any(os.path.getmtime(d) > os.path.getmtime(out4) for d in deps4):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", root, src4, "-o", out4, "-L", _HERE, "-lpwn_hip",
                               "-Wl,-rpath,$ORIGIN/../g2o_frontend_amd"])
    # PwnCloser::processPartition over the GPUs of a node in native code: the mirror + RCCL (broadcast of the flat `current` cloud, all-gather of the
    # match records); the program's own streams and events come from the HIP runtime
    src5 = os.path.join(root, "tools", "pwn_hip_partition_app.cpp")
    out5 = os.path.join(root, "tools", "pwn_hip_partition_app")
    deps5 = [src5] + deps[1:]
    if os.path.exists("/opt/rocm/include/rccl/rccl.h") and (force or (not os.path.exists(out5)) or any(os.path.getmtime(d) > os.path.getmtime(out5) for d in deps5)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I", root, "-I", "/opt/rocm/include", src5, "-o", out5, "-L", _HERE, "-lpwn_hip",
                               "-L", "/opt/rocm/lib", "-lrccl", "-lamdhip64", "-Wl,-rpath,$ORIGIN/../g2o_frontend_amd", "-Wl,-rpath,/opt/rocm/lib"])
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_tools(force="--force" in sys.argv))
