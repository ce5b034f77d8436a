#!/usr/bin/env python3
"""Single-pair latency breakdown: wall time of convert (2 frames) and align (1 pair) against the sum of the kernels' own durations
(hipEvent pairs), to see how much of the latency path is launch / dependency gaps."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
import torch
from g2o_frontend_amd import api, synth
rows, cols = 480, 640
K, conv, alig = bench.conf(rows, cols)
ctx = api.Context(0, rows, cols, 2)
converter, aligner = bench.build_objects(ctx, rows, cols, K, conv, alig)
r, c, _ = synth.make_pair(0, rows, cols, K)
rd = torch.from_numpy(r.view(np.int16)).cuda(); cd = torch.from_numpy(c.view(np.int16)).cuda()
a, b = api.Cloud(ctx, rows * cols), api.Cloud(ctx, rows * cols)
names = ["unproject", "integral", "integral_rows", "integral_cols", "stats", "project_cur", "project_ref", "corr_linearize", "solve"]
for prof in (False, True):
    ctx.set_profiling(prof)
    tc, ta, st = [], [], {k: 0.0 for k in names}
    for it in range(30):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        converter.computeBatch([a, b], [rd, cd], raw_scale=0.001)
        t1 = time.perf_counter()
        if prof:
            for k in names[:5]: st[k] += ctx.stage_ms(k)[0]
        t1b = time.perf_counter()
        aligner.alignBatch([a], [b])
        t2 = time.perf_counter()
        if prof:
            for k in names[5:]: st[k] += ctx.stage_ms(k)[0]
        if it >= 5: tc.append(t1 - t0); ta.append(t2 - t1b)
    print("profiling", prof, "convert(2 frames) %.3f ms  align(1 pair) %.3f ms" % (np.median(tc) * 1e3, np.median(ta) * 1e3))
    if prof:
        print("  kernel ms per call:", {k: round(v / 30, 4) for k, v in st.items()}, " convert sum %.3f align sum %.3f" % (sum(st[k] for k in names[:5]) / 30, sum(st[k] for k in names[5:]) / 30))
ctx.close()
