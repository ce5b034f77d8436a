#!/usr/bin/env python3
"""Single alignments and tracker frames with the index-image shortcut on and off (pwn_hip_debug_set_index_shortcut), alternating, medians.
python tools/exp_index_shortcut.py"""
import concurrent.futures as cf
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bench
from g2o_frontend_amd import api, synth

rows, cols, K = 480, 640, synth.K_VGA
NF = 100
poses = synth.trajectory_sweep(9, NF)
with cf.ThreadPoolExecutor(16) as ex:
    fr = list(ex.map(lambda k: synth.render_depth_mm(9, poses[k], rows, cols, K, hole_stream=k), range(NF)))
_, conv, alig = bench.conf(rows, cols)
ctx = api.Context(0, rows, cols, 2)
converter, al = bench.build_objects(ctx, rows, cols, K, conv, alig)
alproj = api.PinholePointProjector(); alproj.setMinDistance(alig["min_distance"]); alproj.setMaxDistance(alig["max_distance"])
al.setProjector(alproj)
tracker = api.PwnTracker(al, converter); tracker.setScale(1)
Km = np.array([[K[0], 0, K[2]], [0, K[1], K[3]], [0, 0, 1]], np.float32)
I = np.eye(4, dtype=np.float32)
frames = [ctx.DepthImage_convert_16UC1_to_32FC1(f) for f in fr]
for d in frames[:5]:
    tracker.processFrame(d, I, Km)
res = {0: dict(align=[], frame=[]), 1: dict(align=[], frame=[])}
for rep in range(8):
    for enabled in (0, 1):
        ctx.check(ctx._L.pwn_hip_debug_set_index_shortcut(ctx.h, enabled))
        tracker.init()
        t = time.perf_counter()
        for d in frames:
            tracker.processFrame(d, I, Km)
        res[enabled]["frame"].append((time.perf_counter() - t) / NF * 1e6)
        t = time.perf_counter()
        for k in range(NF):
            al.align()
        res[enabled]["align"].append((time.perf_counter() - t) / NF * 1e6)
for enabled in (0, 1):
    print(f"index shortcut {'on ' if enabled else 'off'}: align {np.median(res[enabled]['align']):.0f} us (runs {' '.join('%.0f' % v for v in res[enabled]['align'])}); "
          f"tracker frame {np.median(res[enabled]['frame']):.0f} us (runs {' '.join('%.0f' % v for v in res[enabled]['frame'])})", flush=True)
