#!/usr/bin/env python3
"""The tracker leg of bench.py on its own, and where the look-ahead's time goes: frames from pageable memory, from page-locked memory,
and with every host-side part timed.  python tools/exp_tracker_bench_leg.py"""
import concurrent.futures as cf
import json
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bench
from g2o_frontend_amd import api, synth

rows, cols, K = 480, 640, synth.K_VGA
NF = 200
poses = synth.trajectory_sweep(9, NF)
with cf.ThreadPoolExecutor(16) as ex:
    fr = list(ex.map(lambda k: synth.render_depth_mm(9, poses[k], rows, cols, K, hole_stream=k), range(NF)))
print(json.dumps(bench.run_tracker(0, fr, poses)), flush=True)

_, conv, alig = bench.conf(rows, cols)
ctx = api.Context(0, rows, cols, 2)
converter, al = bench.build_objects(ctx, rows, cols, K, conv, alig)
alproj = api.PinholePointProjector(); alproj.setMinDistance(alig["min_distance"]); alproj.setMaxDistance(alig["max_distance"])
al.setProjector(alproj)
tracker = api.PwnTracker(al, converter); tracker.setScale(1)
Km = np.array([[K[0], 0, K[2]], [0, K[1], K[3]], [0, 0, 1]], np.float32)
I = np.eye(4, dtype=np.float32)
pageable = [ctx.DepthImage_convert_16UC1_to_32FC1(f) for f in fr]
pinned = []
for f in pageable:
    a = api.pinned_empty(f.shape, np.float32); a[...] = f; pinned.append(a)
resident = [ctx.upload(f) for f in pageable[:NF]]


def loop(frames, ahead, label):
    tracker.init()
    t = time.perf_counter()
    for k, d in enumerate(frames):
        tracker.processFrame(d, I, Km, nextDepthImage=(frames[k + 1] if ahead and k + 1 < len(frames) else None))
    dt = (time.perf_counter() - t) / len(frames)
    print(f"{label}: {dt * 1e3:.3f} ms/frame = {1 / dt:.0f} frames/s", flush=True)


for rep in range(2):
    loop(pageable, False, "plain, pageable frames")
    loop(pageable, True, "look-ahead, pageable frames")
    loop(pinned, False, "plain, page-locked frames")
    loop(pinned, True, "look-ahead, page-locked frames")

# the parts of a look-ahead frame on the host
c0 = tracker.makeCloud(Km, I, pageable[0])[0]; c1 = tracker.makeCloud(Km, I, pageable[1])[0]
al.setReferenceCloud(c0); al.setCurrentCloud(c1); al.setInitialGuess(I)
al.align()
for frames, name in ((pageable, "pageable"), (pinned, "page-locked")):
    tb = te = ta = 0.0
    for k in range(NF):
        t0 = time.perf_counter(); t = tracker.makeCloudBegin(Km, I, frames[k]); t1 = time.perf_counter()
        al.align(); t2 = time.perf_counter()
        tracker.makeCloudEnd(t); t3 = time.perf_counter()
        tb += t1 - t0; ta += t2 - t1; te += t3 - t2
    print(f"{name}: makeCloudBegin {tb / NF * 1e6:.0f} us, align next to the conversion {ta / NF * 1e6:.0f} us, makeCloudEnd {te / NF * 1e6:.0f} us", flush=True)
t0 = time.perf_counter()
for k in range(NF):
    al.align()
print(f"align alone {(time.perf_counter() - t0) / NF * 1e6:.0f} us", flush=True)
for enabled in (0, 1, 0, 1):
    ctx.check(ctx._L.pwn_hip_debug_set_index_shortcut(ctx.h, enabled))
    al.align()
    t0 = time.perf_counter()
    for k in range(NF):
        al.align()
    print(f"align alone, index shortcut {'on' if enabled else 'off'}: {(time.perf_counter() - t0) / NF * 1e6:.0f} us", flush=True)
    loop(pageable, False, f"plain tracker, index shortcut {'on' if enabled else 'off'}")
