"""Where does the time of host-frame batches go?  (run on the GPU box)  python tools/exp_h2d.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from g2o_frontend_amd import api, synth

rows, cols = 480, 640
K, conv, alig = bench.conf(rows, cols)
n = 256
ctx = api.Context(0, rows, cols, 128)
ctx.set_subbatch(64, 64)
converter, aligner = bench.build_objects(ctx, rows, cols, K, conv, alig)
base = [synth.make_pair(s, rows, cols, K)[0] for s in range(8)]
frames = [base[i % 8] for i in range(n)]
clouds = [api.Cloud(ctx, rows * cols) for _ in range(n)]
resident = [ctx.upload(f) for f in frames]
sep = [api.pinned_empty((rows, cols), np.uint16) for _ in range(n)]
block = api.pinned_empty((n, rows, cols), np.uint16)
for i, f in enumerate(frames):
    sep[i][...] = f; block[i] = f
pageable = [np.ascontiguousarray(f).copy() for f in frames]


def timeit(name, inputs, reps=10, streams=2):
    ctx.set_concurrency(streams)
    prep = converter.batchHandles(clouds, inputs)
    converter.computeBatch(clouds, inputs, raw_scale=0.001, prepared=prep)
    ctx.synchronize(); t = time.perf_counter()
    for _ in range(reps):
        converter.computeBatch(clouds, inputs, raw_scale=0.001, prepared=prep)
    ctx.synchronize()
    print(f"{name:34s} {(time.perf_counter() - t) / reps * 1e3:7.3f} ms per 256 frames", flush=True)


for streams in (2, 1):
    print("streams", streams)
    timeit("resident", resident, streams=streams)
    timeit("pinned, separate buffers", sep, streams=streams)
    timeit("pinned, one block", [block[i] for i in range(n)], streams=streams)
    timeit("pageable", pageable, streams=streams)
# the bare copies
dst = resident
L = ctx._L
import ctypes as C
t = time.perf_counter()
for i in range(n):
    ctx.check(L.pwn_hip_copy(ctx.h, C.c_void_p(dst[i].data_ptr()), sep[i].ctypes.data_as(C.c_void_p), sep[i].nbytes))
print(f"256 synchronous pinned copies        {(time.perf_counter() - t) * 1e3:7.3f} ms")
big = ctx.upload(np.zeros((n, rows, cols), np.uint16))
t = time.perf_counter()
ctx.check(L.pwn_hip_copy(ctx.h, C.c_void_p(big.data_ptr()), block.ctypes.data_as(C.c_void_p), block.nbytes))
dt = time.perf_counter() - t
print(f"one 157 MB pinned copy               {dt * 1e3:7.3f} ms = {block.nbytes / dt / 1e9:.1f} GB/s")
