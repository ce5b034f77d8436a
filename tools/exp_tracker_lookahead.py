#!/usr/bin/env python3
"""Where does a tracker frame's time go, and what would converting frame k+1 while frame k is aligned buy?
python tools/exp_tracker_lookahead.py"""
import concurrent.futures as cf
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from g2o_frontend_amd import api, synth

rows, cols, K = 480, 640, synth.K_VGA
NF = 200
poses = synth.trajectory_sweep(9, NF)
with cf.ThreadPoolExecutor(16) as ex:
    frames_mm = list(ex.map(lambda k: synth.render_depth_mm(9, poses[k], rows, cols, K, hole_stream=k), range(NF)))
_, conv, alig = bench.conf(rows, cols)
Km = np.array([[K[0], 0, K[2]], [0, K[1], K[3]], [0, 0, 1]], np.float32)
I = np.eye(4, dtype=np.float32)


def objects(ctx):
    converter, al = bench.build_objects(ctx, rows, cols, K, conv, alig)
    alproj = api.PinholePointProjector(); alproj.setMinDistance(alig["min_distance"]); alproj.setMaxDistance(alig["max_distance"])
    al.setProjector(alproj)
    return converter, al


ctx = api.Context(0, rows, cols, 2)
converter, al = objects(ctx)
frames = [ctx.DepthImage_convert_16UC1_to_32FC1(f) for f in frames_mm]
tracker = api.PwnTracker(al, converter); tracker.setScale(1)
for d in frames[:3]:
    tracker.processFrame(d, I, Km)


def run(tr, label):
    tr.init()
    t = time.perf_counter()
    for d in frames:
        tr.processFrame(d, I, Km)
    dt = (time.perf_counter() - t) / NF
    print(f"{label}: {dt * 1e3:.3f} ms/frame = {1 / dt:.0f} frames/s; keyframes {tr.numKeyframes()}", flush=True)
    return tr.globalT().copy()


T0 = run(tracker, "plain")
T0 = run(tracker, "plain")
# parts
t = time.perf_counter()
for d in frames:
    c = tracker.makeCloud(Km, I, d)[0]
print(f"makeCloud alone (upload + scale + convert + count read-back): {(time.perf_counter() - t) / NF * 1e3:.3f} ms", flush=True)
c0 = tracker.makeCloud(Km, I, frames[0])[0]; c1 = tracker.makeCloud(Km, I, frames[1])[0]
al.setReferenceCloud(c0); al.setCurrentCloud(c1); al.setInitialGuess(I)
al.align()
t = time.perf_counter()
for _ in range(NF):
    al.align()
print(f"align alone: {(time.perf_counter() - t) / NF * 1e3:.3f} ms", flush=True)

# look-ahead: a second context and a host thread convert frame k+1 while frame k is aligned
ctxB = api.Context(0, rows, cols, 2)
converterB, _ = objects(ctxB)


class Ahead(api.PwnTracker):
    def __init__(self, aligner, converter, converter_b, ctx_b):
        super().__init__(aligner, converter)
        self.helper = api.PwnMatcherBase(aligner, converter_b); self.ctx_b = ctx_b
        self.pool = cf.ThreadPoolExecutor(1); self.pending = None

    def setScale(self, s):
        super().setScale(s); self.helper.setScale(s)

    def prefetch(self, depth, offset, Kc):
        self.pending = (depth, self.pool.submit(self.helper.makeCloud, Kc, offset, depth, self.ctx_b))

    def makeCloud(self, Kc, offset, depth, ctx=None):
        if self.pending is not None and self.pending[0] is depth:
            fut = self.pending[1]; self.pending = None
            return fut.result()
        return super().makeCloud(Kc, offset, depth, ctx)


sys.setswitchinterval(1e-4)
ahead = Ahead(al, converter, converterB, ctxB); ahead.setScale(1)
for rep in range(2):
    ahead.init()
    t = time.perf_counter()
    ahead.prefetch(frames[0], I, Km)
    for k, d in enumerate(frames):
        # the order a streaming caller would use: hand over frame k+1, then process frame k
        nxt = frames[k + 1] if k + 1 < NF else None
        cur_pending = ahead.pending
        r = None
        # processFrame(k) takes the cloud prefetched for k; frame k+1 is submitted as soon as k's cloud has been taken
        class _Hook: pass
        cloud_k = ahead.makeCloud  # noqa
        def mc(Kc, offset, depth, ctx=None, _orig=api.PwnTracker.makeCloud):
            out = Ahead.makeCloud(ahead, Kc, offset, depth, ctx)
            if nxt is not None:
                ahead.prefetch(nxt, I, Km)
            return out
        ahead.makeCloud = mc
        ahead.processFrame(d, I, Km)
        del ahead.makeCloud
    dt = (time.perf_counter() - t) / NF
    T1 = ahead.globalT().copy()
    print(f"look-ahead (second context + host thread): {dt * 1e3:.3f} ms/frame = {1 / dt:.0f} frames/s; keyframes {ahead.numKeyframes()}; "
          f"globalT bitwise equal to the plain run: {np.array_equal(T0.view(np.uint32), T1.view(np.uint32))}", flush=True)
