"""PwnMatcherBase::makeCloud / matchClouds (pwn_tracker/pwn_matcher_base.cpp:57-183) and the
acceptance rule of PwnCloser::matchFrames (pwn_tracker/pwn_closer.cpp:56-58,138-141)."""
import numpy as np
import pytest

from conftest import case_params, make_depth_pair


# ------------------------------------------------------------------------------------------------ oracle (CPU)
def test_match_score_semantics_cpu(oracle):
    """The reference's `abs(cur-ref) & mask` is a bitwise AND of float words with bits(255.0f) = 0x437F0000:
    differences whose exponent loses a bit are halved / collapsed (4..7 mm -> 2..3.5, 512..1023 mm -> [2,4))."""
    FM = np.finfo(np.float32).max
    base = 1.5
    deltas = np.array([0, 1, 2, 3, 4, 5, 7, 8, 15, 16, 31, 32, 49, 50, 51, 63, 64, 100, 127, 128, 200, 255, 256, 300, 511, 512, 600, 1023, 1024, 2000], np.float32)
    cur = np.full(len(deltas) + 3, base, np.float32)
    ref = np.concatenate([base + deltas / 1000.0, [FM, 0.0, base]]).astype(np.float32)
    cur[-1] = FM                                                     # empty current pixel
    s = oracle.match_score(ref, cur, 50.0)
    cu = (np.float32(1000.0) * cur[:-1]).astype(np.uint16); ru = np.where(ref[:-1] < FM, (np.float32(1000.0) * np.minimum(ref[:-1], 60.0)).astype(np.uint16), 0)
    m = (cu > 0) & (ru > 0)
    ad = np.abs(cu.astype(np.float32) - ru.astype(np.float32))
    d = (ad.view(np.uint32) & np.uint32(0x437F0000)).view(np.float32)
    assert s["image_nonZeros"] == int(m.sum()) == len(deltas)
    assert s["image_inliers"] == int((m & (d < 50)).sum()) and s["image_outliers"] == s["image_nonZeros"] - s["image_inliers"]
    assert abs(s["image_reprojectionDistance"] - d[m].sum() / m.sum()) < 1e-4
    # the documented quirks
    q = dict(zip(deltas.tolist(), d[: len(deltas)].tolist()))
    assert q[5.0] == 2.5 and q[7.0] == 3.5 and q[16.0] == 8.0 and q[50.0] == 50.0 and q[100.0] == 50.0 and q[600.0] < 4.0 and q[2.0] == 2.0
    assert q[1.0] < 1e-30


# ------------------------------------------------------------------------------------------------ GPU
pytest_gpu = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from g2o_frontend_amd import api
    c = api.Context(device=0, max_rows=480, max_cols=640, max_batch=4, omega_storage="exact9")
    yield c
    c.close()


def _matcher(ctx, name):
    from g2o_frontend_amd import api
    from test_gpu_parity import gpu_objects
    proj, converter, aligner = gpu_objects(ctx, name)
    return api.PwnMatcherBase(aligner, converter), proj, converter, aligner


@pytest_gpu
def test_match_score_quirk_cases_gpu(ctx, oracle):
    """Hand-built depth images whose per-column differences hit every exponent case of the bitwise-and quirk."""
    from g2o_frontend_amd import api
    rows, cols, K = 24, 32, (30.0, 30.0, 15.5, 11.5)
    deltas = [0, 1, 2, 3, 4, 5, 7, 8, 15, 16, 31, 32, 49, 50, 51, 63, 64, 100, 127, 128, 200, 255, 256, 300, 511, 512, 600, 1023, 1024, 2000, 0, 0]
    cur = np.full((rows, cols), 1.5, np.float32)
    ref = (cur + np.array(deltas, np.float32)[None, :] / 1000.0).astype(np.float32)
    ref[3, :] = 0.0; cur[5, 4:9] = 0.0                                # holes on either side
    proj = api.PinholePointProjector(); proj.setCameraMatrix([[K[0], 0, K[2]], [0, K[1], K[3]], [0, 0, 1]])
    proj.setMinDistance(0.5); proj.setMaxDistance(5.0); proj.setImageSize(rows, cols)
    cr, cc = api.Cloud(ctx, rows * cols), api.Cloud(ctx, rows * cols)
    proj.unProject(cr, ref); proj.unProject(cc, cur)
    f = api.CorrespondenceFinder(); f.setImageSize(rows, cols)
    lin = api.Linearizer(); al = api.Aligner(ctx)
    al.setProjector(proj); al.setLinearizer(lin); al.setCorrespondenceFinder(f); al.setOuterIterations(1)
    al.setReferenceCloud(cr); al.setCurrentCloud(cc)
    al.align(images=True)
    rd, cd = f.referenceDepthImage(), f.currentDepthImage()
    assert np.array_equal(rd[ref > 0], ref[ref > 0]) and np.array_equal(cd[cur > 0], cur[cur > 0])     # projection of unprojection is exact
    from g2o_frontend_amd._lib import MatchResult
    import ctypes as C
    m = MatchResult()
    ctx.check(ctx._L.pwn_hip_match_score(ctx.h, 50.0, C.byref(m)))
    o = oracle.match_score(rd, cd, 50.0)
    assert (m.image_non_zeros, m.image_inliers, m.image_outliers) == (o["image_nonZeros"], o["image_inliers"], o["image_outliers"])
    assert abs(m.image_reprojection_distance - o["image_reprojectionDistance"]) <= 1e-5 * o["image_reprojectionDistance"]
    assert 0 < m.image_outliers < m.image_non_zeros


@pytest_gpu
@pytest.mark.parametrize("name,seed", [("small", 1), ("vga", 0)])
def test_matchClouds_matches_oracle_chain(ctx, oracle, name, seed):
    """makeCloud (DepthImage_scale + convert at 1/scale) and matchClouds (align + score) against the same chain in the oracle."""
    from g2o_frontend_amd import api, synth
    matcher, proj, converter, aligner = _matcher(ctx, "small")           # scale-4 stats parameters
    rows, cols, K, _, _ = case_params("vga")
    _, _, _, conv, alig = case_params("small")
    matcher.setScale(4)
    ref_mm, cur_mm, Ttrue = synth.make_pair(seed + 20, rows, cols, K)
    ref, cur = oracle.convert_16u_to_32f(ref_mm), oracle.convert_16u_to_32f(cur_mm)
    Km = np.array([[K[0], 0, K[2]], [0, K[1], K[3]], [0, 0, 1]], np.float32)
    I = np.eye(4, dtype=np.float32)
    cf, r, c, Ks = matcher.makeCloud(Km, I, ref)
    ct, _, _, _ = matcher.makeCloud(Km, I, cur)
    assert (r, c) == (120, 160) and Ks[0, 0] == np.float32(525.0) * np.float32(0.25) and Ks[2, 2] == 1
    # oracle chain
    K4 = (float(Ks[0, 0]), float(Ks[1, 1]), float(Ks[0, 2]), float(Ks[1, 2]))
    cp = oracle.converter_params(K=K4, **conv)
    ocf, _, _ = oracle.convert(cp, oracle.depth_scale(ref, 4)); oct_, _, _ = oracle.convert(cp, oracle.depth_scale(cur, 4))
    a, b = ocf.arrays(), cf.arrays()
    for k in ("points", "normals", "curvature", "omega_p"):
        assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)) or np.array_equal(a[k], b[k]), k
    guess = np.eye(4); guess[2, 3] = 0.3                                 # z of the guess is zeroed by matchClouds (.cpp:114)
    res = matcher.matchClouds(cf, ct, I, I, Km, rows, cols, guess)
    ap = oracle.aligner_params(120, 160, K=K4, accumulate_fp64=1, **alig)
    o = oracle.align(ap, ocf, oct_, images=True)
    os_ = oracle.match_score(o["ref_depth"], o["cur_depth"], 50.0)
    assert np.abs(res["transform"] - o["T"]).max() < 2e-5 and abs(res["cloud_inliers"] - o["inliers"]) <= 4
    assert abs(res["image_nonZeros"] - os_["image_nonZeros"]) <= 8 and abs(res["image_inliers"] - os_["image_inliers"]) <= 8
    assert abs(res["image_reprojectionDistance"] - os_["image_reprojectionDistance"]) <= 2e-2 * os_["image_reprojectionDistance"] + 1e-3
    assert np.array_equal(res["informationMatrix"], np.eye(6) * 100)
    # score of the GPU's own finder images is exact against the oracle scoring of those same images
    f = aligner.correspondenceFinder()
    aligner.align(images=True)
    ex = oracle.match_score(f.referenceDepthImage(), f.currentDepthImage(), 50.0)
    assert (res["image_nonZeros"], res["image_inliers"], res["image_outliers"]) == (ex["image_nonZeros"], ex["image_inliers"], ex["image_outliers"])
    assert abs(res["image_reprojectionDistance"] - ex["image_reprojectionDistance"]) <= 1e-3 * ex["image_reprojectionDistance"]
    acc = api.PwnCloserAcceptance()
    assert acc.accept(res) == (not (res["image_nonZeros"] < 3000 or res["image_outliers"] > 100 or res["image_inliers"] < 1000))


@pytest_gpu
def test_matchCloudsBatch_equals_single(ctx, oracle):
    from g2o_frontend_amd import synth
    matcher, proj, converter, aligner = _matcher(ctx, "small")
    matcher.setScale(4)
    rows, cols, K, _, _ = case_params("vga")
    Km = np.array([[K[0], 0, K[2]], [0, K[1], K[3]], [0, 0, 1]], np.float32)
    I = np.eye(4, dtype=np.float32)
    cur_mm = synth.render_depth_mm(31, np.eye(4), rows, cols, K)
    current, _, _, _ = matcher.makeCloud(Km, I, oracle.convert_16u_to_32f(cur_mm))
    others, guesses = [], []
    for k in range(3):                                                   # same `current`, different `other` (pwn_closer.cpp:92-111)
        pose = synth.pair_pose(100 + k)
        mm = synth.render_depth_mm(31, pose, rows, cols, K, hole_stream=k + 1)
        c, _, _, _ = matcher.makeCloud(Km, I, oracle.convert_16u_to_32f(mm))
        others.append(c); guesses.append(np.eye(4))
    # identity guesses: the batch path takes the clouds' own index images for the current projection and the first reference projection and
    # reads the current depth image of the score off the cloud; non-identity guesses: only the current side.  Both must equal the single calls
    # (which project everything) bit for bit.
    moved = [synth.v2t(np.array([0.01 * (k + 1), -0.005, 0.0, 0.002, -0.001 * k, 0.001])) for k in range(3)]
    for gs in (guesses, moved):
        single = [matcher.matchClouds(current, o, I, I, Km, rows, cols, g) for o, g in zip(others, gs)]
        batch = matcher.matchCloudsBatch([current] * 3, others, I, I, Km, rows, cols, gs)
        for s, b in zip(single, batch):
            assert np.array_equal(s["transform"], b["transform"])
            for k in ("image_nonZeros", "image_outliers", "image_inliers", "cloud_inliers"):
                assert s[k] == b[k], k
            assert s["image_reprojectionDistance"] == b["image_reprojectionDistance"]
    # the same with the roles swapped: many currents against one reference
    single = [matcher.matchClouds(o, current, I, I, Km, rows, cols, g) for o, g in zip(others, moved)]
    batch = matcher.matchCloudsBatch(others, [current] * 3, I, I, Km, rows, cols, moved)
    for s, b in zip(single, batch):
        assert np.array_equal(s["transform"], b["transform"]) and s["image_nonZeros"] == b["image_nonZeros"] and s["image_inliers"] == b["image_inliers"]
        assert s["image_reprojectionDistance"] == b["image_reprojectionDistance"]


@pytest_gpu
def test_finder_images_end_with_their_clouds(ctx, oracle):
    """The aligner's z-buffers keep point indices; depth images and the matchClouds score are recomputed from the clouds' points.  Once a cloud
    of the last alignment has new content or is gone, pwn_hip_align_images / pwn_hip_match_score refuse instead of reading freed or foreign points."""
    import ctypes as C
    from g2o_frontend_amd import api
    from g2o_frontend_amd._lib import MatchResult, PwnHipError
    from test_gpu_parity import gpu_objects
    rows, cols, K, _, _ = case_params("small")
    ref, cur, _, _, _ = make_depth_pair("small", 4)
    _, converter, aligner = gpu_objects(ctx, "small")
    gref, gcur = api.Cloud(ctx, rows * cols), api.Cloud(ctx, rows * cols)
    converter.compute(gref, ref); converter.compute(gcur, cur)
    aligner.setReferenceCloud(gref); aligner.setCurrentCloud(gcur)
    aligner.align(images=True)
    m = MatchResult()
    ctx.check(ctx._L.pwn_hip_match_score(ctx.h, 50.0, C.byref(m)))
    assert m.image_non_zeros > 1000
    converter.compute(gcur, ref)                                   # the current cloud gets new content
    with pytest.raises(PwnHipError) as e:
        ctx.check(ctx._L.pwn_hip_match_score(ctx.h, 50.0, C.byref(m)))
    assert e.value.code == 1
    aligner.align()
    ctx.check(ctx._L.pwn_hip_match_score(ctx.h, 50.0, C.byref(m)))   # valid again after the next alignment
    ctx.check(ctx._L.pwn_hip_cloud_destroy(ctx.h, gref.h)); gref.h = None
    with pytest.raises(PwnHipError):
        aligner.align.__self__.ctx.check(ctx._L.pwn_hip_align_images(ctx.h, None, None, None, None))
