#!/usr/bin/env python3
"""Scene-maintenance stage on one MI355X: per-call time of the sensor-noise Gaussians, Cloud::add,
Merger::merge and VoxelCalculator::compute on VGA clouds, with the compulsory bytes each call moves (formulae below) and the
CPU oracle timed on the same inputs.  Prints one JSON line.

Compulsory bytes (M = points of a frame, n = points of the scene before the call, k = after; N = pixels of the merger's view):
  gaussians : 4N depth in, 100M out (24 floats + flags)
  add       : 268M in (point 16, normal 16, omega_p 36, omega_n 36, stats 64, Gaussian 100) + 268M out
  merge     : project 16n + 8N; classify 16n + 8n (z-buffer word) + 32n normals + 4n; accumulate ~100n; compaction 268(n + k)
  voxelize  : 16n keys, 268k gather (sort temporaries not counted)
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402


def main():
    from g2o_frontend_amd import api, synth
    from oracle import oracle as O      # CPU baseline leg only
    rows, cols = 480, 640
    N = rows * cols
    K, conv, alig = bench.conf(rows, cols)
    n_frames = 4
    poses = synth.trajectory(5, n_frames)
    depths = [O.convert_16u_to_32f(synth.render_depth_mm(5, poses[k], rows, cols, K, hole_stream=k)) for k in range(n_frames)]
    rel = [(np.linalg.inv(poses[0]) @ poses[k]).astype(np.float32) for k in range(n_frames)]
    ctx = api.Context(0, rows, cols, 2)
    converter, _ = bench.build_objects(ctx, rows, cols, K, conv, alig)
    clouds = [api.Cloud(ctx, N) for _ in range(n_frames)]
    tm = {k: [] for k in ("gaussians", "add", "merge", "voxelize")}
    by = {k: [] for k in tm}
    reps = 5
    for rep in range(reps + 1):
        scene = api.Cloud(ctx, n_frames * N)
        merger = api.Merger(); merger.setDepthImageConverter(converter); merger.setImageSize(rows, cols)
        for k in range(n_frames):
            converter.compute(clouds[k], depths[k], keep_stats=True, images=False)
            p = converter.params(None)
            import ctypes as C
            ctx.synchronize(); a = time.perf_counter()
            ctx.check(ctx._L.pwn_hip_cloud_gaussians(ctx.h, C.byref(p), depths[k].ctypes.data_as(C.c_void_p), rows, cols, clouds[k].h, 0.075, 0.1))
            ctx.synchronize(); b = time.perf_counter()
            M = clouds[k].size()
            n0 = scene.size()
            scene.add(clouds[k], rel[k])
            ctx.synchronize(); c = time.perf_counter()
            n1 = scene.size()
            kk = merger.merge(scene, rel[k])
            ctx.synchronize(); d = time.perf_counter()
            if rep > 0:
                tm["gaussians"].append(b - a); by["gaussians"].append(4 * N + 100 * M)     # includes the H2D copy of the depth image
                tm["add"].append(c - b); by["add"].append(2 * 268 * M)
                tm["merge"].append(d - c); by["merge"].append(16 * n1 + 8 * N + 60 * n1 + 100 * n1 + 268 * (n1 + kk))
        n1 = scene.size()
        ctx.synchronize(); a = time.perf_counter()
        kv = api.VoxelCalculator().compute(scene, 0.02)
        ctx.synchronize(); b = time.perf_counter()
        if rep > 0:
            tm["voxelize"].append(b - a); by["voxelize"].append(16 * n1 + 268 * kv)
        scene_final, voxels = n1, kv
        del scene
    # CPU oracle on the same sequence (single thread)
    O.set_num_threads(1); O.set_gaussians(True)
    cp = O.converter_params(K=K, **conv)
    oc = [O.convert(cp, d)[0] for d in depths]
    cpu = {}
    s = O.Cloud(); ta = tmg = 0.0
    for k in range(n_frames):
        a = time.perf_counter(); s.add(oc[k], rel[k]); b = time.perf_counter()
        O.merge(s, K, rel[k], conv["min_distance"], conv["max_distance"], rows, cols); c = time.perf_counter()
        ta += b - a; tmg += c - b
    a = time.perf_counter(); O.voxelize(s, 0.02, False); b = time.perf_counter()
    cpu = {"add_ms": ta / n_frames * 1e3, "merge_ms": tmg / n_frames * 1e3, "voxelize_ms": (b - a) * 1e3, "threads": 1,
           "scene_points": len(s)}
    O.set_gaussians(False)
    out = {"stage": "scene maintenance, 4 VGA frames appended and merged one by one", "scene_points_after_merges": scene_final, "voxels_2cm": voxels,
           "gpu": {k: {"ms": float(np.median(v)) * 1e3, "GBps": float(np.median(np.array(by[k]) / np.array(v))) / 1e9} for k, v in tm.items()},
           "cpu_oracle": cpu}
    print(json.dumps(out))
    ctx.close()


if __name__ == "__main__":
    main()
