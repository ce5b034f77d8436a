"""cProfile of the Python tracker loop (where does a frame's 0.5 ms go on the host side?)  python tools/exp_tracker_profile.py"""
import cProfile, os, pstats, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from g2o_frontend_amd import api, synth

rows, cols, K = 480, 640, synth.K_VGA
poses = synth.trajectory_sweep(9, 60)
frames_mm = [synth.render_depth_mm(9, poses[k], rows, cols, K, hole_stream=k) for k in range(60)]
_, conv, alig = bench.conf(rows, cols)
ctx = api.Context(0, rows, cols, 2)
converter, al = bench.build_objects(ctx, rows, cols, K, conv, alig)
alproj = api.PinholePointProjector(); alproj.setMinDistance(alig["min_distance"]); alproj.setMaxDistance(alig["max_distance"])
al.setProjector(alproj)
tracker = api.PwnTracker(al, converter); tracker.setScale(1)
Km = np.array([[K[0], 0, K[2]], [0, K[1], K[3]], [0, 0, 1]], np.float32)
I = np.eye(4, dtype=np.float32)
frames = [ctx.DepthImage_convert_16UC1_to_32FC1(f) for f in frames_mm]
for d in frames[:3]:
    tracker.processFrame(d, I, Km)
tracker.init()
t = time.perf_counter()
for d in frames:
    tracker.processFrame(d, I, Km)
print("plain: %.3f ms/frame" % ((time.perf_counter() - t) / len(frames) * 1e3))
tracker.init()
pr = cProfile.Profile(); pr.enable()
for d in frames:
    tracker.processFrame(d, I, Km)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
