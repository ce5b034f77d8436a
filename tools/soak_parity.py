#!/usr/bin/env python3
"""Parity soak: many seeded pairs, random sensor offsets and converter settings -- converter arrays bit for bit, first-iteration
counters exactly, chi2 from the same iterate to 1e-5, free-running pose; scene add + merge bit for bit.  Prints a summary line per case
and a final JSON; exits non-zero on the first mismatch.  (The committed tests cover fixed seeds; this is the wide net.)

Round 6: per case also the DECISIONS the reference's callers take from a free-running alignment, on both sides -- PwnCloser::matchFrames' acceptance
(pwn_tracker/pwn_closer.cpp:138-141: image_nonZeros < 3000 || image_outliers > 100 || image_inliers < 1000 -> rejected, on the matchClouds score of the
finder's depth images, pwn_matcher_base.cpp:153-182) and PwnTracker::processFrame's two (pwn_tracker.cpp:146-151 transform found = inliers > 0;
:162-164 new key frame = inliers / (rows * cols) < 0.4).  Free-running poses differ in their last digits (summation order); what a caller does with
them should not.  A flip is reported with the numbers on both sides, not asserted."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def bits(a):
    a = np.ascontiguousarray(a)
    if a.dtype != np.float32:
        return a
    a = a.copy(); a[a == 0] = 0
    return a.view(np.uint32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--vga", type=int, default=16)
    ap.add_argument("--small", type=int, default=60)
    ap.add_argument("--seed0", type=int, default=1000, help="first seed (the committed runs of rounds 3-5 used 1000; another value = other scenes, offsets and settings)")
    ap.add_argument("--omega-storage", choices=("exact9", "sym6"), default="exact9",
                    help="sym6: the clouds keep the upper triangle of the point information matrices -- everything else bit for bit, the mirrored triangle "
                         "within 1e-6 |Omega_p|, in scenes (where T Omega T^t reads the mirrored matrix) within 4e-6")
    ap.add_argument("--check-fallback", action="store_true",
                    help="every alignment once more with the projection's settle loop forced to give up (0 rounds: the call is repeated with the two-pass "
                         "projection): poses, traces and counters must be the same bits")
    args = ap.parse_args()
    sym = args.omega_storage == "sym6"
    from conftest import case_params
    from g2o_frontend_amd import api, synth
    from oracle import oracle as O
    from test_gpu_parity import gpu_objects
    ctx = api.Context(0, 480, 640, 16, omega_storage=args.omega_storage)
    if sym:
        from test_omega_sym6 import compare_clouds_sym6
    rng = np.random.default_rng(2024 + args.seed0 - 1000)
    stats = dict(cases=0, worst_chi2_rel=0.0, worst_pose=0.0, points=0, merged=0, decisions=0, closer_accepted=0, tracker_new_keyframe=0, worst_score_diff=0,
                 worst_inliers_diff=0)
    flips = []
    import ctypes as C
    from g2o_frontend_amd._lib import MatchResult
    for name, count in (("small", args.small), ("vga", args.vga)):
        rows, cols, K, conv0, alig = case_params(name)
        kept = []          # (aligner params key, gref, gcur, single result) of the default-configuration cases: re-run as one batch below
        for seed in range(args.seed0, args.seed0 + count):
            conv = dict(conv0)
            offset = None
            if seed % 3 == 1:      # random sensor mounting
                q = rng.uniform(-0.3, 0.3, 3); t = rng.uniform(-0.2, 0.2, 3)
                offset = synth.v2t(np.concatenate([t, q])).astype(np.float32)
            if seed % 4 == 2:      # other window / threshold settings
                conv["min_image_radius"] = int(rng.integers(2, 12)); conv["max_image_radius"] = conv["min_image_radius"] + int(rng.integers(1, 20))
                conv["min_points"] = int(rng.integers(5, 80)); conv["stats_curvature_threshold"] = float(rng.uniform(0.01, 0.3))
            noisy = seed % 5 == 3      # sensor-like z-noise: many non-flat / rejected neighbourhoods, every branch of the eigensolver
            ref_mm, cur_mm, Ttrue = synth.make_pair(seed, rows, cols, K, holes=float(rng.uniform(0.0, 0.2)), noise=noisy)
            ref, cur = O.convert_16u_to_32f(ref_mm), O.convert_16u_to_32f(cur_mm)
            O.set_gaussians(True)
            cp = O.converter_params(K=K, sensor_offset=offset, **conv)
            oref, oidx, oitv = O.convert(cp, ref); ocur, _, _ = O.convert(cp, cur)
            O.set_gaussians(False)
            proj, converter, aligner = gpu_objects(ctx, name, sensor_offset=offset)
            st = converter._stats
            st.setMinImageRadius(conv["min_image_radius"]); st.setMaxImageRadius(conv["max_image_radius"]); st.setMinPoints(conv["min_points"])
            st.setCurvatureThreshold(conv["stats_curvature_threshold"])
            gref, gcur = api.Cloud(ctx, rows * cols), api.Cloud(ctx, rows * cols)
            converter.compute(gref, ref, sensorOffset=offset, keep_stats=True, gaussians=True)
            assert np.array_equal(converter.indexImage(), oidx) and np.array_equal(converter.intervalImage(), oitv), (name, seed, "images")
            converter.compute(gcur, cur, sensorOffset=offset, keep_stats=True, gaussians=True)
            for o, g in ((oref, gref), (ocur, gcur)):
                oa, ga = o.arrays(stats=True), g.arrays(stats=True)
                if sym:
                    compare_clouds_sym6(oa, ga)
                for k in oa:
                    if sym and k == "omega_p":
                        continue
                    assert np.array_equal(bits(oa[k]), bits(ga[k])), (name, seed, k)
                og, gg = o.gaussians(), g.gaussians()
                assert np.array_equal(bits(og["cov"]), bits(gg["cov"])) and np.array_equal(bits(og["mean"]), bits(gg["mean"])), (name, seed, "gaussians")
            apar = O.aligner_params(rows, cols, K=K, accumulate_fp64=1, reference_sensor_offset=offset, current_sensor_offset=offset, **alig)
            o = O.align(apar, oref, ocur, images=True)
            aligner.setReferenceCloud(gref); aligner.setCurrentCloud(gcur)
            g = aligner.align()
            # the callers' decisions, both sides
            gm = MatchResult()
            ctx.check(ctx._L.pwn_hip_match_score(ctx.h, 50.0, C.byref(gm)))
            om = O.match_score(o["ref_depth"], o["cur_depth"], 50.0)
            dec_o = (not (om["image_nonZeros"] < 3000 or om["image_outliers"] > 100 or om["image_inliers"] < 1000), o["inliers"] > 0, o["inliers"] / float(rows * cols) < 0.4)
            dec_g = (not (gm.image_non_zeros < 3000 or gm.image_outliers > 100 or gm.image_inliers < 1000), g["inliers"] > 0, g["inliers"] / float(rows * cols) < 0.4)
            stats["decisions"] += 3
            stats["closer_accepted"] += int(dec_o[0]); stats["tracker_new_keyframe"] += int(dec_o[2])
            stats["worst_score_diff"] = max(stats["worst_score_diff"], abs(om["image_nonZeros"] - gm.image_non_zeros), abs(om["image_outliers"] - gm.image_outliers),
                                            abs(om["image_inliers"] - gm.image_inliers))
            stats["worst_inliers_diff"] = max(stats["worst_inliers_diff"], abs(int(o["inliers"]) - int(g["inliers"])))
            if dec_o != dec_g:
                flips.append(dict(case=name, seed=seed, oracle=dict(score=om, inliers=int(o["inliers"]), decisions=dec_o),
                                  gpu=dict(score=dict(image_nonZeros=gm.image_non_zeros, image_outliers=gm.image_outliers, image_inliers=gm.image_inliers),
                                           inliers=int(g["inliers"]), decisions=dec_g)))
            it0 = o["iterations"][0]
            assert (int(g["K"][0]), int(g["C"][0]), int(g["iter_inliers"][0])) == (it0["K"], it0["C"], it0["inliers"]), (name, seed, "counters")
            if args.check_fallback:
                before = C.c_int(0); after = C.c_int(0)
                ctx.check(ctx._L.pwn_hip_debug_projection_fallbacks(ctx.h, C.byref(before)))
                ctx.check(ctx._L.pwn_hip_debug_set_settle_guard(ctx.h, 0))
                g2 = aligner.align()
                ctx.check(ctx._L.pwn_hip_debug_set_settle_guard(ctx.h, -1))
                ctx.check(ctx._L.pwn_hip_debug_projection_fallbacks(ctx.h, C.byref(after)))
                for key in ("T", "chi2", "K", "C", "iter_inliers"):
                    assert np.array_equal(bits(np.asarray(g[key])), bits(np.asarray(g2[key]))), (name, seed, "two-pass projection", key)
                stats["fallback_checked"] = stats.get("fallback_checked", 0) + 1
                stats["fallback_repeats"] = stats.get("fallback_repeats", 0) + (after.value - before.value)
            if offset is None and conv == conv0 and len(kept) < 16:
                kept.append((gref, gcur, g))
            rel = abs(float(g["chi2"][0]) - it0["chi2_fp64"]) / max(it0["chi2_fp64"], 1e-30)
            assert rel <= 1e-5, (name, seed, "chi2", rel)
            pose = float(np.abs(g["T"] - o["T"]).max())
            # free-running distances are properties of the pair (one correspondence that flips at the finder's thresholds moves the iterates apart): bars wide
            # enough for what four seed sets have shown (VGA clean worst 6.7e-5, 120 x 160 clean 2.6e-4, noisy 1.5e-4); the 1e-5 contract is the teacher-forced one above
            assert pose <= ((1e-4 if name == "vga" else 5e-4) if not noisy else 2e-3), (name, seed, "pose", pose)
            # scene: add both views, merge in the first view
            oscene = O.Cloud(); gscene = api.Cloud(ctx, 2 * rows * cols)
            oscene.add(oref, np.eye(4)); gscene.add(gref, np.eye(4)); oscene.add(ocur, o["T"]); gscene.add(gcur, o["T"])
            so = np.eye(4, dtype=np.float32) if offset is None else offset
            merger = api.Merger(); merger.setDepthImageConverter(converter); merger.setImageSize(rows, cols)
            ok, ocol = O.merge(oscene, K, so, conv["min_distance"], conv["max_distance"], rows, cols)
            assert merger.merge(gscene, so) == ok and np.array_equal(merger.collapsedIndices(), ocol), (name, seed, "merge")
            oa, ga = oscene.arrays(stats=True), gscene.arrays(stats=True)
            for k in oa:
                if sym and k == "omega_p":
                    fin = np.isfinite(oa[k]).all(1) & np.isfinite(ga[k]).all(1)
                    sc = np.abs(oa[k][fin]).max(1, keepdims=True)
                    assert np.array_equal(np.isfinite(oa[k]), np.isfinite(ga[k])) and (np.abs(oa[k][fin] - ga[k][fin]) <= 4e-6 * sc).all(), (name, seed, "scene", k)
                    continue
                assert np.array_equal(bits(oa[k]), bits(ga[k])), (name, seed, "scene", k)
            stats["cases"] += 1; stats["worst_chi2_rel"] = max(stats["worst_chi2_rel"], rel); stats["worst_pose"] = max(stats["worst_pose"], pose)
            stats["points"] += len(oref) + len(ocur); stats["merged"] += int(((ocol >= 0) & (ocol != np.arange(len(ocol)))).sum())
            print(f"{name} seed {seed}: M {len(oref)}/{len(ocur)} offset {offset is not None} noise {noisy} chi2 rel {rel:.1e} pose {pose:.1e} merged {ok} | closer "
                  f"{'accept' if dec_o[0] else 'reject'}/{'accept' if dec_g[0] else 'reject'} (nz {om['image_nonZeros']}/{gm.image_non_zeros} out {om['image_outliers']}/{gm.image_outliers} "
                  f"inl {om['image_inliers']}/{gm.image_inliers}) tracker found {int(dec_o[1])}/{int(dec_g[1])} newkey {int(dec_o[2])}/{int(dec_g[2])} (inliers {o['inliers']}/{g['inliers']})"
                  f"{'  <-- DECISION FLIP' if dec_o != dec_g else ''}", flush=True)
        # the batch path (own index images instead of two of the eleven projections, two streams) against the single alignments: bitwise
        if len(kept) > 1:
            _, _, aligner = gpu_objects(ctx, name)
            for sub in (64, 3):
                ctx.set_subbatch(sub, sub)
                res = aligner.alignBatch([k[0] for k in kept], [k[1] for k in kept])
                for r, (_, _, g) in zip(res, kept):
                    assert np.array_equal(r["T"].view(np.uint32), g["T"].view(np.uint32)) and np.array_equal(r["chi2"].view(np.uint32), g["chi2"].view(np.uint32)), (name, "batch", sub)
                    assert np.array_equal(r["C"], g["C"]) and np.array_equal(r["K"], g["K"])
            ctx.set_subbatch(64, 64)
            stats["batch_checked"] = stats.get("batch_checked", 0) + len(kept)
            print(f"{name}: batch of {len(kept)} pairs bitwise equal to the single alignments", flush=True)
        del kept
    stats["omega_storage"] = args.omega_storage
    stats["decision_flips"] = len(flips)
    for f in flips:
        print("DECISION FLIP", json.dumps(f, default=lambda x: bool(x) if isinstance(x, (np.bool_,)) else float(x)))
    print(json.dumps(stats))
    ctx.close()


if __name__ == "__main__":
    main()
