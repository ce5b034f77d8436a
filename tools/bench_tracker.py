#!/usr/bin/env python3
"""BASELINE configs[2]: pwn_tracker sequential odometry on a 200-frame synthetic VGA depth stream, 1 GPU.
Frame k+1 depends on the key-cloud decision of frame k, so this is a latency benchmark (one convert + one align per
frame, strictly sequential: pwn_tracker/pwn_tracker.cpp:106-215).  Prints one JSON line."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from g2o_frontend_amd import api, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=200)
    ap.add_argument("--scale", type=int, default=1, help="matcher scale (the reference app uses 4, BASELINE asks for full VGA = 1)")
    args = ap.parse_args()
    rows, cols, K = 480, 640, synth.K_VGA
    ctx = api.Context(0, rows, cols, 2)
    proj = api.PinholePointProjector(); proj.setMinDistance(0.5); proj.setMaxDistance(4.5)
    st = api.StatsCalculatorIntegralImage(); st.setCurvatureThreshold(0.2)
    if args.scale >= 4:
        st.setMinImageRadius(3); st.setMaxImageRadius(6); st.setMinPoints(10)
    converter = api.DepthImageConverterIntegralImage(proj, st, api.PointInformationMatrixCalculator(), api.NormalInformationMatrixCalculator())
    alproj = api.PinholePointProjector(); alproj.setMinDistance(0.5); alproj.setMaxDistance(4.5)
    f = api.CorrespondenceFinder(); f.setInlierDistanceThreshold(1.0 if args.scale == 1 else 0.5); f.setInlierNormalAngularThreshold(0.95)
    lin = api.Linearizer(); al = api.Aligner(ctx)
    al.setProjector(alproj); al.setLinearizer(lin); al.setCorrespondenceFinder(f)
    tracker = api.PwnTracker(al, converter); tracker.setScale(args.scale)
    Km = np.array([[K[0], 0, K[2]], [0, K[1], K[3]], [0, 0, 1]], np.float32)
    I = np.eye(4, dtype=np.float32)
    poses = synth.trajectory(9, args.frames)
    frames = [ctx.DepthImage_convert_16UC1_to_32FC1(synth.render_depth_mm(9, poses[k], rows, cols, K, hole_stream=k)) for k in range(args.frames)]
    tracker.processFrame(frames[0], I, Km); tracker.processFrame(frames[1], I, Km); tracker.init()      # warm-up
    t0 = time.perf_counter()
    for d in frames:
        r = tracker.processFrame(d, I, Km)
    dt = time.perf_counter() - t0
    true = np.linalg.inv(poses[0]) @ poses[-1]
    err = float(np.abs(tracker.globalT()[:3, 3] - true[:3, 3]).max())
    print(json.dumps({"metric": "tracker frames/s (sequential odometry, 640x480 stream)", "value": args.frames / dt, "unit": "frames/s",
                      "frames": args.frames, "scale": args.scale, "ms_per_frame": dt / args.frames * 1e3, "keyframes": tracker.numKeyframes(),
                      "final_translation_error_m": err, "input": "host float32 frames (PCIe upload inside the timed loop)"}))
    ctx.close()


if __name__ == "__main__":
    main()
