#!/usr/bin/env python3
"""Converter wall time per call for small batches (n VGA frames resident in HBM), for the library named by PWN_HIP_LIB."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from g2o_frontend_amd import api, synth
from test_gpu_parity import gpu_objects

rows, cols = 480, 640
ctx = api.Context(0, rows, cols, 128)
_, conv, _ = gpu_objects(ctx, "vga")
frames = [torch.from_numpy(synth.make_pair(s, rows, cols, synth.K_VGA)[0].view(np.int16)).cuda() for s in range(4)]
out = []
for n in (1, 2, 3, 4, 6, 8, 16, 32, 64):
    clouds = [api.Cloud(ctx, rows * cols) for _ in range(n)]
    src = [frames[i % 4] for i in range(n)]
    prep = conv.batchHandles(clouds, src)
    for _ in range(5):
        conv.computeBatch(clouds, src, raw_scale=0.001, prepared=prep)
    t = time.perf_counter()
    reps = 30
    for _ in range(reps):
        conv.computeBatch(clouds, src, raw_scale=0.001, prepared=prep)
    out.append((n, round((time.perf_counter() - t) / reps * 1e3, 3)))
print(os.environ.get("PWN_HIP_LIB", "default"), out)
