"""Host cost of one batch alignment call around its device work: python tools/exp_call_overhead.py
(n = 0: the fixed Python + C overhead; n = 128 with zero iterations: + per-pair descriptors, uploads and one wait)"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from g2o_frontend_amd import api, synth

rows, cols = 480, 640
K, conv, alig = bench.conf(rows, cols)
ctx = api.Context(0, rows, cols, 128, omega_storage="sym6")
converter, aligner = bench.build_objects(ctx, rows, cols, K, conv, alig)
alproj = api.PinholePointProjector(); alproj.setMinDistance(alig["min_distance"]); alproj.setMaxDistance(alig["max_distance"])
aligner.setProjector(alproj)
matcher = api.PwnMatcherBase(aligner, converter); matcher.setScale(1)
Km = np.array([[K[0], 0, K[2]], [0, K[1], K[3]], [0, 0, 1]], np.float32); I = np.eye(4, dtype=np.float32)
ids = list(range(128))
mm = synth.render_depth_mm(bench.PARTITION_SCENE, np.eye(4), rows, cols, K, hole_stream=0)
clouds = [api.Cloud(ctx, rows * cols) for _ in range(129)]
converter.computeBatch(clouds, [mm] * 129, raw_scale=0.001)
guesses = bench.partition_guesses(ids)
rec = ctx.upload(np.zeros((128, api.MATCH_RECORD_FLOATS), np.float32))


def timeit(f, n=50):
    f(); f()
    t = time.perf_counter()
    for _ in range(n):
        f()
    return (time.perf_counter() - t) / n * 1e6


for outer in (0, 1, 10):
    aligner.setOuterIterations(outer)
    for n in (1, 16, 128):
        prep = matcher.matchHandles([clouds[0]] * n, clouds[1:1 + n], guesses[:n])
        us = timeit(lambda: matcher.matchCloudsBatchRecords(None, None, I, I, Km, rows, cols, rec, pair_ids=np.asarray(ids[:n], np.int32), want_results=False, prepared=prep))
        print(f"outer {outer:2d} n {n:3d}: {us:9.1f} us per call")
ctx.close()
