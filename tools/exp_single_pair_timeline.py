#!/usr/bin/env python3
"""One VGA pair (convert 2 resident frames + align), repeated; run under `rocprofv3 --kernel-trace --memory-copy-trace --output-format csv` to get
the device timeline of one repetition (tools/summarize_timeline.py reads the CSVs).  python tools/exp_single_pair_timeline.py [reps]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bench
from g2o_frontend_amd import api, synth

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rows, cols, K = 480, 640, synth.K_VGA
_, conv, alig = bench.conf(rows, cols)
ctx = api.Context(0, rows, cols, 2)
converter, al = bench.build_objects(ctx, rows, cols, K, conv, alig)
r, c, _ = synth.make_pair(0, rows, cols, K)
fr, fc = ctx.upload(r), ctx.upload(c)
ref, cur = api.Cloud(ctx, rows * cols), api.Cloud(ctx, rows * cols)
lat = []
for k in range(reps):
    t = time.perf_counter()
    converter.computeBatch([ref, cur], [fr, fc], raw_scale=0.001)
    t1 = time.perf_counter()
    al.alignBatch([ref], [cur])
    t2 = time.perf_counter()
    lat.append(((t1 - t) * 1e6, (t2 - t1) * 1e6))
lat = np.array(lat[5:])
print("median: convert 2 frames %.0f us, align %.0f us, pair %.0f us" % (np.median(lat[:, 0]), np.median(lat[:, 1]), np.median(lat.sum(1))), flush=True)
ctx.close()
