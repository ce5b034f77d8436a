#!/usr/bin/env python3
"""One case of tools/soak_parity.py looked at closely: replays the soak's random stream up to the case, then prints the free-running trace of the GPU
next to the oracle's (K_i, C_i, inliers_i, chi2_i), re-runs every iteration of the oracle's trace from the oracle's own iterate (teacher-forced) and
reports where the two free runs part.  Usage: tools/soak_case.py --seed0 7000 --size small --seed 7094 [--omega-storage sym6]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed0", type=int, default=1000)
    ap.add_argument("--size", choices=("small", "vga"), default="small")
    ap.add_argument("--seed", type=int, required=True)
    ap.add_argument("--small", type=int, default=60, help="the soak run's --small (the vga cases draw from the stream after them)")
    ap.add_argument("--omega-storage", choices=("exact9", "sym6"), default="exact9")
    args = ap.parse_args()
    from conftest import case_params
    from g2o_frontend_amd import api, synth
    from oracle import oracle as O
    from test_gpu_parity import gpu_objects
    rng = np.random.default_rng(2024 + args.seed0 - 1000)
    found = None
    for name, count in (("small", args.small), ("vga", 10 ** 9)):      # the same draws in the same order as soak_parity.py
        rows, cols, K, conv0, alig = case_params(name)
        for seed in range(args.seed0, args.seed0 + count):
            conv = dict(conv0); offset = None
            if seed % 3 == 1:
                q = rng.uniform(-0.3, 0.3, 3); t = rng.uniform(-0.2, 0.2, 3)
                offset = synth.v2t(np.concatenate([t, q])).astype(np.float32)
            if seed % 4 == 2:
                conv["min_image_radius"] = int(rng.integers(2, 12)); conv["max_image_radius"] = conv["min_image_radius"] + int(rng.integers(1, 20))
                conv["min_points"] = int(rng.integers(5, 80)); conv["stats_curvature_threshold"] = float(rng.uniform(0.01, 0.3))
            holes = float(rng.uniform(0.0, 0.2))
            if name == args.size and seed == args.seed:
                found = (rows, cols, K, conv, alig, offset, holes, seed % 5 == 3)
                break
        if found or name == args.size:
            break
    assert found, "case not in the stream"
    rows, cols, K, conv, alig, offset, holes, noisy = found
    print(f"{args.size} seed {args.seed}: holes {holes:.3f} noise {noisy} offset {offset is not None} converter {conv}")
    ctx = api.Context(0, rows, cols, 4, omega_storage=args.omega_storage)
    ref_mm, cur_mm, _ = synth.make_pair(args.seed, rows, cols, K, holes=holes, noise=noisy)
    ref, cur = O.convert_16u_to_32f(ref_mm), O.convert_16u_to_32f(cur_mm)
    cp = O.converter_params(K=K, sensor_offset=offset, **conv)
    oref, _, _ = O.convert(cp, ref); ocur, _, _ = O.convert(cp, cur)
    proj, converter, aligner = gpu_objects(ctx, args.size, sensor_offset=offset)
    st = converter._stats
    st.setMinImageRadius(conv["min_image_radius"]); st.setMaxImageRadius(conv["max_image_radius"]); st.setMinPoints(conv["min_points"])
    st.setCurvatureThreshold(conv["stats_curvature_threshold"])
    gref, gcur = api.Cloud(ctx, rows * cols), api.Cloud(ctx, rows * cols)
    converter.compute(gref, ref, sensorOffset=offset); converter.compute(gcur, cur, sensorOffset=offset)
    apar = O.aligner_params(rows, cols, K=K, accumulate_fp64=1, reference_sensor_offset=offset, current_sensor_offset=offset, **alig)
    o = O.align(apar, oref, ocur)
    aligner.setReferenceCloud(gref); aligner.setCurrentCloud(gcur)
    g = aligner.align()
    print("free-running traces (GPU | oracle):")
    first = None
    for i, it in enumerate(o["iterations"]):
        gk, gc, gi = int(g["K"][i]), int(g["C"][i]), int(g["iter_inliers"][i])
        rel = abs(float(g["chi2"][i]) - it["chi2_fp64"]) / max(it["chi2_fp64"], 1e-30)
        same = (gk, gc, gi) == (it["K"], it["C"], it["inliers"])
        if not same and first is None:
            first = i
        print(f"  it {i}: K {gk} | {it['K']}   C {gc} | {it['C']}   inliers {gi} | {it['inliers']}   chi2 {float(g['chi2'][i]):.6f} | {it['chi2_fp64']:.6f}  rel {rel:.1e}{'' if same else '   <- counters differ'}")
    print(f"final pose: max |dT| {float(np.abs(g['T'] - o['T']).max()):.2e}; first iteration with different counters: {first}")
    # every iteration again from the oracle's own iterate
    aligner.setOuterIterations(1)
    worst = 0.0; exact = True
    for i, it in enumerate(o["iterations"]):
        aligner.setInitialGuess(it["T_before"])
        t = aligner.align()
        same = (int(t["K"][0]), int(t["C"][0]), int(t["iter_inliers"][0])) == (it["K"], it["C"], it["inliers"])
        exact = exact and same
        worst = max(worst, abs(float(t["chi2"][0]) - it["chi2_fp64"]) / max(it["chi2_fp64"], 1e-30))
    print(f"teacher-forced (the oracle's iterate on both sides): worst chi2 rel {worst:.1e}, counters exact in every iteration: {exact}")
    ctx.close()


if __name__ == "__main__":
    main()
