"""Converter-only workload for rocprofv3 --pmc passes (one process, no children): 256 resident VGA frames through computeBatch,
serial launches of 64 frames.  python tools/pmc_convert.py [reps]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from g2o_frontend_amd import api, synth
rows, cols = 480, 640
K, conv, alig = bench.conf(rows, cols)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
ctx = api.Context(0, rows, cols, 128); ctx.set_subbatch(64, 64); ctx.set_concurrency(1)
converter, aligner = bench.build_objects(ctx, rows, cols, K, conv, alig)
base = [synth.make_pair(s, rows, cols, K)[0] for s in range(4)]
frames = [ctx.upload(base[i % 4]) for i in range(256)]
clouds = [api.Cloud(ctx, rows * cols) for _ in range(256)]
prep = converter.batchHandles(clouds, frames)
for _ in range(reps):
    converter.computeBatch(clouds, None, raw_scale=0.001, prepared=prep)
ctx.synchronize()
print("done")
