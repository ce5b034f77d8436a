"""PinholePointProjector::project (pwn_core/pinholepointprojector.cpp:33-66) under pathological pixel collisions.

The aligner's projection kernel settles points of one projection that meet in a pixel with a compare-and-swap loop (z32_settle,
csrc/pwn_kernels.h).  The loop is bounded; a thread that runs out of rounds raises the call's fault word and the library repeats the call with
a two-pass projection that needs no loop.  Either way the finder's images are the oracle's -- never a silent difference:
  * a VGA cloud projected into a 24 x 32 thumbnail (400 points per pixel),
  * a cloud of 2^16 + 4464 points on one viewing ray (every point in ONE pixel), with equal depths for the tie rule,
  * the fallback forced (0 rounds) on single alignments, batches, the prior path and the one-submission step: same bits as the default path.
"""
import ctypes as C

import numpy as np
import pytest

from conftest import case_params, make_depth_pair
from test_gpu_parity import gpu_objects, oracle_params

pytestmark = pytest.mark.gpu


def _fallbacks(ctx):
    n = C.c_int(0)
    ctx.check(ctx._L.pwn_hip_debug_projection_fallbacks(ctx.h, C.byref(n)))
    return n.value


def _set_guard(ctx, rounds):
    ctx.check(ctx._L.pwn_hip_debug_set_settle_guard(ctx.h, int(rounds)))


def _images_equal(finder, o):
    assert np.array_equal(finder.referenceIndexImage(), o["ref_index"])
    assert np.array_equal(finder.currentIndexImage(), o["cur_index"])
    assert np.array_equal(finder.referenceDepthImage().view(np.uint32), o["ref_depth"].view(np.uint32))
    assert np.array_equal(finder.currentDepthImage().view(np.uint32), o["cur_depth"].view(np.uint32))


@pytest.mark.parametrize("guard", [None, 2, 0])
def test_vga_cloud_into_a_24x32_thumbnail(oracle, guard):
    """307 200-pixel clouds projected into 768 pixels: ~400 points of one projection per word.  guard = None is the product setting; 2 and 0
    make threads give up early, so the repeat with the two-pass projection is what produces the images."""
    from g2o_frontend_amd import api, synth
    rows, cols, K, conv, alig = case_params("vga")
    ref, cur, _, _, _ = make_depth_pair("vga", 5)
    ctx = api.Context(0, rows, cols, 4)
    try:
        _, converter, aligner = gpu_objects(ctx, "vga")
        gref, gcur = api.Cloud(ctx, rows * cols), api.Cloud(ctx, rows * cols)
        converter.compute(gref, ref); converter.compute(gcur, cur)
        cp, _ = oracle_params(oracle, "vga")
        oref, _, _ = oracle.convert(cp, ref); ocur, _, _ = oracle.convert(cp, cur)
        shrink = 20
        r2, c2 = rows // shrink, cols // shrink
        K2 = synth.scaled_K(K, shrink)
        aligner.projector().setCameraMatrix([[K2[0], 0, K2[2]], [0, K2[1], K2[3]], [0, 0, 1]]); aligner.projector().setImageSize(r2, c2)
        aligner.correspondenceFinder().setImageSize(r2, c2)
        aligner.setReferenceCloud(gref); aligner.setCurrentCloud(gcur)
        aligner.setOuterIterations(1)
        g0 = synth.v2t(np.array([0.02, -0.01, 0.01, 0.01, -0.01, 0.02])).astype(np.float32)
        aligner.setInitialGuess(g0)
        if guard is not None:
            _set_guard(ctx, guard)
        before = _fallbacks(ctx)
        aligner.align(images=True)
        ap = oracle.aligner_params(r2, c2, K=K2, initial_guess=g0, accumulate_fp64=1, **dict(alig, outer_iterations=1))
        o = oracle.align(ap, oref, ocur, images=True)
        assert int((o["ref_index"] >= 0).sum()) <= r2 * c2 and len(oref) > 100 * r2 * c2
        _images_equal(aligner.correspondenceFinder(), o)
        if guard == 0:
            assert _fallbacks(ctx) == before + 1          # every collision gave up: the call was repeated
    finally:
        ctx.close()


def _ray_cloud(n, seed, ties=True):
    """n points on the optical axis (all project to the principal point), depths in [1, 4] m in random order; with ties: the nearest depth
    occurs at several indices (the reference keeps the first: strict '>')."""
    rng = np.random.default_rng(seed)
    z = rng.uniform(1.0, 4.0, n).astype(np.float32)
    if ties:
        zmin = np.float32(0.9)
        z[rng.choice(n, 7, replace=False)] = zmin
    pts = np.zeros((n, 4), np.float32); pts[:, 2] = z; pts[:, 3] = 1.0
    nrm = np.zeros((n, 4), np.float32); nrm[:, 2] = -1.0
    curv = np.full(n, 0.01, np.float32)
    om = np.zeros((n, 16), np.float32); om[:, 0] = om[:, 5] = om[:, 10] = 1.0
    return pts, nrm, curv, om, om.copy()


@pytest.mark.parametrize("guard", [None, 0])
def test_seventy_thousand_points_in_one_pixel(oracle, guard):
    from g2o_frontend_amd import api
    rows, cols, K, conv, alig = case_params("small")
    n = (1 << 16) + 4464
    ctx = api.Context(0, rows, cols, 4)
    try:
        _, _, aligner = gpu_objects(ctx, "small")
        ra, ca = _ray_cloud(n, 1), _ray_cloud(n, 2)
        gref, gcur = api.Cloud(ctx, n), api.Cloud(ctx, n)
        gref.upload(*ra); gcur.upload(*ca)
        oref, ocur = oracle.Cloud.from_arrays(*ra), oracle.Cloud.from_arrays(*ca)
        aligner.setReferenceCloud(gref); aligner.setCurrentCloud(gcur)
        aligner.setOuterIterations(1)
        if guard is not None:
            _set_guard(ctx, guard)
        before = _fallbacks(ctx)
        aligner.align(images=True)
        ap = oracle.aligner_params(rows, cols, K=K, accumulate_fp64=1, **dict(alig, outer_iterations=1))
        o = oracle.align(ap, oref, ocur, images=True)
        assert int((o["ref_index"] >= 0).sum()) == 1 and int((o["cur_index"] >= 0).sum()) == 1
        # the winner is the FIRST of the equally near points
        assert o["ref_index"].max() == int(np.flatnonzero(ra[0][:, 2] == ra[0][:, 2].min())[0])
        _images_equal(aligner.correspondenceFinder(), o)
        if guard == 0:
            assert _fallbacks(ctx) == before + 1
        # a batch of 9 of the same pair (four points per thread) -- same words
        single = aligner.align()
        for b in aligner.alignBatch([gref] * 9, [gcur] * 9):
            assert np.array_equal(b["T"].view(np.uint32), single["T"].view(np.uint32)) and np.array_equal(b["K"], single["K"]) and np.array_equal(b["C"], single["C"])
    finally:
        ctx.close()


def test_forced_fallback_gives_the_bits_of_the_default_path(oracle):
    """0 rounds: every collision gives up, every call below runs a second time with the two-pass projection.  Non-identity guess and a 2x
    shrunk image, so that both clouds are really projected and collide (4 points per pixel).  Single alignment, batch, match batch with records,
    the prior path and the one-submission step: results bitwise those of the default path."""
    from g2o_frontend_amd import api, synth
    rows, cols, K, conv, alig = case_params("small")
    ref, cur, _, ref_mm, cur_mm = make_depth_pair("small", 4)
    g0 = synth.v2t(np.array([0.02, -0.01, 0.01, 0.01, -0.01, 0.02])).astype(np.float32)

    def run(forced):
        ctx = api.Context(0, rows, cols, 32)
        try:
            _, converter, aligner = gpu_objects(ctx, "small")
            gref, gcur = api.Cloud(ctx, rows * cols), api.Cloud(ctx, rows * cols)
            converter.compute(gref, ref); converter.compute(gcur, cur)
            r2, c2 = rows // 2, cols // 2
            K2 = synth.scaled_K(K, 2)
            aligner.projector().setCameraMatrix([[K2[0], 0, K2[2]], [0, K2[1], K2[3]], [0, 0, 1]]); aligner.projector().setImageSize(r2, c2)
            aligner.correspondenceFinder().setImageSize(r2, c2)
            aligner.setReferenceCloud(gref); aligner.setCurrentCloud(gcur); aligner.setInitialGuess(g0)
            if forced:
                _set_guard(ctx, 0)
            out = {}
            f0 = _fallbacks(ctx)
            out["single"] = aligner.align(images=True)
            f = aligner.correspondenceFinder()
            out["images"] = (f.referenceIndexImage().copy(), f.currentIndexImage().copy(), f.referenceDepthImage().copy(), f.currentDepthImage().copy())
            f1 = _fallbacks(ctx)
            out["batch"] = aligner.alignBatch([gref] * 20, [gcur] * 20, initialGuesses=[g0] * 20)
            f2 = _fallbacks(ctx)
            aligner.clearPriors()
            aligner.addAbsolutePrior(np.eye(4, dtype=np.float32), g0, np.eye(6, dtype=np.float32) * 10.0)
            out["prior"] = aligner.align()
            aligner.clearPriors()
            f3 = _fallbacks(ctx)
            # the one-submission step (20 pairs from raw frames, full resolution, identity guess): the repeat converts the frames again
            _, converter2, aligner2 = gpu_objects(ctx, "small")
            refs = [api.Cloud(ctx, rows * cols) for _ in range(20)]; curs = [api.Cloud(ctx, rows * cols) for _ in range(20)]
            rec = np.zeros((20, api.RECORD_FLOATS), np.float32)
            out["step"] = np.array(aligner2.convertAlignBatch(converter2, refs, curs, [ref_mm] * 20, [cur_mm] * 20, records=rec))
            out["step_records"] = rec
            out["step_cloud"] = curs[7].arrays()
            f4 = _fallbacks(ctx)
            out["fallbacks"] = (f1 - f0, f2 - f1, f3 - f2, f4 - f3)
            return out
        finally:
            ctx.close()

    a, b = run(False), run(True)
    assert a["fallbacks"] == (0, 0, 0, 0) and b["fallbacks"] == (1, 1, 1, 1)
    assert a["step"]["T"].tobytes() == b["step"]["T"].tobytes() and a["step_records"].tobytes() == b["step_records"].tobytes()
    for k, v in a["step_cloud"].items():
        assert np.array_equal(v.view(np.uint32), b["step_cloud"][k].view(np.uint32)), k
    for k in ("single", "prior"):
        for field in ("T", "chi2", "K", "C", "iter_inliers"):
            assert np.array_equal(np.asarray(a[k][field]).view(np.uint32), np.asarray(b[k][field]).view(np.uint32)), (k, field)
    for x, y in zip(a["images"], b["images"]):
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32))
    for x, y in zip(a["batch"], b["batch"]):
        assert np.array_equal(x["T"].view(np.uint32), y["T"].view(np.uint32)) and np.array_equal(x["chi2"].view(np.uint32), y["chi2"].view(np.uint32))
