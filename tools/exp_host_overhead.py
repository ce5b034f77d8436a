"""Host-side cost of the batch calls: alignBatch with 0 outer iterations (descriptors, state upload, read-back, sync -- no kernels of the loop),
per-call wall time of the real calls next to the sum of their kernels' stage times.  python tools/exp_host_overhead.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from g2o_frontend_amd import api, synth

rows, cols = 480, 640
K, conv, alig = bench.conf(rows, cols)
P = 128
ctx = api.Context(0, rows, cols, 128); ctx.set_subbatch(64, 64)
converter, aligner = bench.build_objects(ctx, rows, cols, K, conv, alig)
pairs = [synth.make_pair(s, rows, cols, K) for s in range(4)]
ref = [ctx.upload(pairs[i % 4][0]) for i in range(P)]; cur = [ctx.upload(pairs[i % 4][1]) for i in range(P)]
refs = [api.Cloud(ctx, rows * cols) for _ in range(P)]; curs = [api.Cloud(ctx, rows * cols) for _ in range(P)]
prep = converter.batchHandles(refs + curs, ref + cur)
import ctypes as C
aprep = ((C.c_void_p * P)(*[c.h for c in refs]), (C.c_void_p * P)(*[c.h for c in curs]), P)


def t(f, n=20):
    f(); ctx.synchronize(); a = time.perf_counter()
    for _ in range(n):
        f()
    ctx.synchronize()
    return (time.perf_counter() - a) / n * 1e3


conv_ms = t(lambda: converter.computeBatch(refs + curs, None, raw_scale=0.001, prepared=prep))
al_ms = t(lambda: aligner.alignBatch(refs, curs, raw=True, prepared=aprep))
aligner.setOuterIterations(0)
al0_ms = t(lambda: aligner.alignBatch(refs, curs, raw=True, prepared=aprep))
aligner.setOuterIterations(1)
al1_ms = t(lambda: aligner.alignBatch(refs, curs, raw=True, prepared=aprep))
aligner.setOuterIterations(10)
ctx.set_concurrency(1); ctx.set_profiling(True)
converter.computeBatch(refs + curs, None, raw_scale=0.001, prepared=prep)
cs = sum(ctx.stage_ms(k)[0] for k in ("unproject", "integral", "stats"))
aligner.alignBatch(refs, curs, raw=True, prepared=aprep)
as_ = sum(ctx.stage_ms(k)[0] for k in ("project_cur", "project_ref", "corr_linearize", "solve"))
print(f"convert 256 frames: call {conv_ms:.3f} ms (two streams); kernels alone, serial {cs:.3f} ms")
print(f"align 128 pairs:   call {al_ms:.3f} ms (two streams); kernels alone, serial {as_:.3f} ms")
print(f"align with 0 outer iterations: {al0_ms:.3f} ms; with 1: {al1_ms:.3f} ms")
