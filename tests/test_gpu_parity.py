"""Parity of the HIP path (through the C-ABI) against the CPU oracle, stage by stage and end to end.

Bars (BASELINE.json north_star):
  * integer / index outputs (index images, interval images, correspondence lists, counts): bit-exact;
  * unprojected points, integral image, projected depth images: bit-exact (same fp32 op order, no FMA);
  * normals / curvature / eigenvalues / Stats / both information matrices: bit-exact as well (the eigensolver's
    three trig calls are evaluated by the same fixed double-precision algorithms on both sides, see oracle/pwn_oracle.cpp g_trig_mode);
  * per-iteration chi2 at the SAME iterate (teacher-forced with the oracle's T_i): |d|/chi2 <= 1e-5 against the
    oracle's fp64-accumulated value, counters K_i / C_i / inliers_i exact;
  * free-running chi2 trace: the iterates differ in the last bits (summation order of H, b), which can move a
    projected point across a pixel boundary; a correspondence entering/leaving the set changes chi2 by its own
    term (typically ~chi2/C, up to ~1e-2*chi2 for a boundary outlier at 120x160).  The strict 1e-5 claim is the
    teacher-forced test.  Free-running, the difference is seed-dependent: FREE_CHI2_RTOL_VGA is the bar at VGA
    (C ~ 2e5; measured worst over 16 seeds: test_free_running_chi2_many_seeds prints it), 1e-2 at 120x160;
  * final SE(3): translation <= 1e-5 m, rotation matrix entries <= 1e-5.
"""
import os

import numpy as np
import pytest

from conftest import case_params, make_depth_pair

pytestmark = pytest.mark.gpu

CHI2_RTOL = 1e-5        # north_star: chi2 within 1e-5 rel of CPU -- enforced from the same iterate (teacher-forced)
# free-running trace at VGA: measured over 16 seeds (test_free_running_chi2_many_seeds, profiles/r03_parity_measured.txt): worst 1.9e-4, 9 of 16
# seeds above 1e-5, every one with 1-3 correspondences more or fewer than the oracle at that iteration (or one swapped); a term at the finder's
# thresholds is 20-40 x the mean term (chi2 ~ 32 000 over ~205 000 terms at convergence).  Final poses agree to 3e-7.
FREE_CHI2_RTOL_VGA = 5e-4
POSE_TTOL = 1e-5        # metres
POSE_RTOL = 1e-5        # rotation-matrix entries (~rad)


@pytest.fixture(scope="module")
def ctx():
    from g2o_frontend_amd import api
    c = api.Context(device=0, max_rows=480, max_cols=640, max_batch=4, omega_storage="exact9")
    yield c
    c.close()


def gpu_objects(ctx, name, sensor_offset=None):
    """Reference-style object graph configured like pwn_simple_aligner.cpp:214-269 for the case."""
    from g2o_frontend_amd import api
    rows, cols, K, conv, alig = case_params(name)
    proj = api.PinholePointProjector()
    proj.setCameraMatrix([[K[0], 0, K[2]], [0, K[1], K[3]], [0, 0, 1]])
    proj.setMinDistance(conv["min_distance"]); proj.setMaxDistance(conv["max_distance"])
    proj.setImageSize(rows, cols)
    stats = api.StatsCalculatorIntegralImage()
    stats.setWorldRadius(conv["world_radius"]); stats.setMinImageRadius(conv["min_image_radius"])
    stats.setMaxImageRadius(conv["max_image_radius"]); stats.setMinPoints(conv["min_points"])
    stats.setCurvatureThreshold(conv["stats_curvature_threshold"])
    pinfo, ninfo = api.PointInformationMatrixCalculator(), api.NormalInformationMatrixCalculator()
    pinfo.setCurvatureThreshold(conv["point_info_curvature_threshold"]); ninfo.setCurvatureThreshold(conv["normal_info_curvature_threshold"])
    converter = api.DepthImageConverterIntegralImage(proj, stats, pinfo, ninfo)
    finder = api.CorrespondenceFinder()
    finder.setInlierDistanceThreshold(alig["inlier_distance_threshold"]); finder.setInlierNormalAngularThreshold(alig["inlier_normal_angular_threshold"])
    finder.setFlatCurvatureThreshold(alig["flat_curvature_threshold"]); finder.setInlierCurvatureRatioThreshold(alig["inlier_curvature_ratio_threshold"])
    finder.setImageSize(rows, cols)
    lin = api.Linearizer(); lin.setInlierMaxChi2(alig["inlier_max_chi2"]); lin.setRobustKernel(alig["robust_kernel"])
    aligner = api.Aligner(ctx)
    aligner.setProjector(proj); aligner.setLinearizer(lin); aligner.setCorrespondenceFinder(finder)
    aligner.setOuterIterations(alig["outer_iterations"]); aligner.setInnerIterations(alig["inner_iterations"])
    if sensor_offset is not None:
        aligner.setSensorOffset(sensor_offset)
    return proj, converter, aligner


def oracle_params(O, name, sensor_offset=None, **akw):
    rows, cols, K, conv, alig = case_params(name)
    cp = O.converter_params(K=K, sensor_offset=sensor_offset, **conv)
    alig = dict(alig); alig.update(akw)
    ap = O.aligner_params(rows, cols, K=K, reference_sensor_offset=sensor_offset, current_sensor_offset=sensor_offset, **alig)
    return cp, ap


def upload(ctx, ocloud):
    from g2o_frontend_amd import api
    a = ocloud.arrays()
    c = api.Cloud(ctx, max(1, len(ocloud)))
    c.upload(a["points"], a["normals"], a["curvature"], a["omega_p"], a["omega_n"])
    return c


# ------------------------------------------------------------------------------------------ input conditioning
def test_depth_conversions_bit_exact(ctx, oracle):
    rng = np.random.default_rng(0)
    raw = rng.integers(0, 65535, size=(120, 160), dtype=np.uint16)
    raw[rng.random(raw.shape) < 0.1] = 0
    ref = oracle.convert_16u_to_32f(raw)
    got = ctx.DepthImage_convert_16UC1_to_32FC1(raw)
    assert np.array_equal(ref.view(np.uint32), got.view(np.uint32))
    f = ref.copy(); f[3, 5] = np.finfo(np.float32).max
    assert np.array_equal(oracle.convert_32f_to_16u(f), ctx.DepthImage_convert_32FC1_to_16UC1(f))
    for step in (1, 2, 4):
        a, b = oracle.depth_scale(ref, step), ctx.DepthImage_scale(ref, step)
        assert a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


# ------------------------------------------------------------------------------------------ converter stages
@pytest.mark.parametrize("name,seed", [("small", 1), ("vga", 0)])
def test_unproject_and_intervals_bit_exact(ctx, oracle, name, seed):
    from g2o_frontend_amd import api
    rows, cols, K, conv, _ = case_params(name)
    depth, _, _, _, _ = make_depth_pair(name, seed)
    cp, _ = oracle_params(oracle, name)
    opts, oidx = oracle.unproject(cp, depth)
    oitv = oracle.project_intervals(cp, depth)
    proj, converter, _ = gpu_objects(ctx, name)
    cloud = api.Cloud(ctx, rows * cols)
    gidx = proj.unProject(cloud, depth)
    gitv = proj.projectIntervals(ctx, depth, conv["world_radius"])
    assert cloud.size() == len(opts)
    assert np.array_equal(oidx, gidx)
    assert np.array_equal(oitv, gitv)
    g = cloud.arrays()
    assert np.array_equal(opts.view(np.uint32), g["points"].view(np.uint32))


@pytest.mark.parametrize("name,seed", [("small", 1), ("vga", 0)])
def test_integral_image_bit_exact(ctx, oracle, name, seed):
    """Sequential-order prefix sums: the fp32 sums must carry the same bits as the CPU scan order."""
    from g2o_frontend_amd import api
    rows, cols, K, conv, _ = case_params(name)
    depth, _, _, _, _ = make_depth_pair(name, seed)
    cp, _ = oracle_params(oracle, name)
    opts, oidx = oracle.unproject(cp, depth)
    oI = oracle.integral_image(oidx, opts)
    proj, _, _ = gpu_objects(ctx, name)
    cloud = api.Cloud(ctx, rows * cols)
    gidx = proj.unProject(cloud, depth)
    gI = api.StatsCalculatorIntegralImage.integralImage(cloud, gidx)
    assert np.array_equal(oI.view(np.uint32), gI.view(np.uint32))


def _compare_clouds(o, g, name):
    """Every field of the converted cloud must carry the oracle's bits."""
    assert len(o["points"]) == len(g["points"])
    for k in ("points", "normals", "curvature", "omega_p", "omega_n"):
        a, b = o[k].reshape(len(o[k]), -1), g[k].reshape(len(g[k]), -1)
        # +0 / -0 are the same number (zero rows of the information matrices are built differently)
        same = (a.view(np.uint32) == b.view(np.uint32)) | ((a == 0) & (b == 0))
        bad = int((~same).any(1).sum())
        assert bad == 0, f"{k}: {bad} of {len(a)} points differ, max |d| = {np.abs(a - b).max():.3e}"
    assert (np.abs(o["normals"][:, :3]).sum(1) > 0).mean() > 0.5, "degenerate test input: no normals"


@pytest.mark.parametrize("name,seed", [("small", 1), ("vga", 0)])
def test_convert_matches_oracle(ctx, oracle, name, seed):
    from g2o_frontend_amd import api
    rows, cols, K, conv, _ = case_params(name)
    depth, _, _, _, _ = make_depth_pair(name, seed)
    cp, _ = oracle_params(oracle, name)
    oc, oidx, oitv = oracle.convert(cp, depth)
    _, converter, _ = gpu_objects(ctx, name)
    cloud = api.Cloud(ctx, rows * cols)
    converter.compute(cloud, depth, keep_stats=True)
    assert np.array_equal(oidx, converter.indexImage())
    assert np.array_equal(oitv, converter.intervalImage())
    o, g = oc.arrays(stats=True), cloud.arrays(stats=True)
    _compare_clouds(o, g, name)
    # Stats: n, eigenvalues, eigenvectors + mean
    assert np.array_equal(o["npoints"], g["npoints"])
    assert np.array_equal(o["eigenvalues"].view(np.uint32), g["eigenvalues"].view(np.uint32))
    ok = o["npoints"] > 0
    assert np.array_equal(o["stats"][ok].view(np.uint32), g["stats"][ok].view(np.uint32))


def test_convert_with_sensor_offset(ctx, oracle):
    """Cloud::transformInPlace fused into the converter (cloud.cpp:173-186)."""
    from g2o_frontend_amd import api, synth
    name = "small"
    rows, cols, K, conv, _ = case_params(name)
    depth, _, _, _, _ = make_depth_pair(name, 3)
    off = synth.v2t(np.array([0.1, -0.05, 0.2, 0.03, -0.02, 0.05])).astype(np.float32)
    cp, _ = oracle_params(oracle, name, sensor_offset=off)
    oc, _, _ = oracle.convert(cp, depth)
    _, converter, _ = gpu_objects(ctx, name)
    cloud = api.Cloud(ctx, rows * cols)
    converter.compute(cloud, depth, sensorOffset=off)
    _compare_clouds(oc.arrays(), cloud.arrays(), name)


def test_convert_empty_and_ragged(ctx, oracle):
    """All-invalid image (zero points) and an image with a single valid pixel per row."""
    from g2o_frontend_amd import api
    name = "small"
    rows, cols, K, conv, _ = case_params(name)
    _, converter, _ = gpu_objects(ctx, name)
    cp, _ = oracle_params(oracle, name)
    cloud = api.Cloud(ctx, rows * cols)
    z = np.zeros((rows, cols), np.float32)
    converter.compute(cloud, z)
    assert cloud.size() == 0 and (converter.indexImage() == -1).all()
    rag = z.copy()
    rag[np.arange(rows), (np.arange(rows) * 7) % cols] = 1.5
    rag[5, :] = 2.0
    oc, oidx, _ = oracle.convert(cp, rag)
    converter.compute(cloud, rag)
    assert cloud.size() == len(oc) and np.array_equal(oidx, converter.indexImage())
    _compare_clouds(oc.arrays(), cloud.arrays(), name)


# ------------------------------------------------------------------------------------------ aligner stages
@pytest.mark.parametrize("name,seed", [("small", 1), ("vga", 0)])
def test_project_bit_exact(ctx, oracle, name, seed):
    """Projector index + depth images are bit-exact, including z-buffer ties (lowest index wins)."""
    from g2o_frontend_amd import synth
    rows, cols, K, conv, _ = case_params(name)
    depth, _, Ttrue, _, _ = make_depth_pair(name, seed)
    cp, _ = oracle_params(oracle, name)
    oc, _, _ = oracle.convert(cp, depth)
    gc = upload(ctx, oc)
    proj, _, _ = gpu_objects(ctx, name)
    for T in (np.eye(4), Ttrue, synth.v2t(np.array([0.3, -0.2, 0.4, 0.1, 0.05, -0.08]))):
        T = T.astype(np.float32)
        oi, od = oracle.project(K, T, conv["min_distance"], conv["max_distance"], rows, cols, oc.arrays()["points"])
        proj.setTransform(T)
        gi, gd = proj.project(gc)
        assert np.array_equal(oi, gi)
        assert np.array_equal(od.view(np.uint32), gd.view(np.uint32))


def test_project_ties_and_out_of_range(ctx, oracle):
    """Hand-built cloud: duplicate points (tie -> lowest index), points behind the camera, outside the image."""
    from g2o_frontend_amd import api
    rows, cols, K = 24, 32, (30.0, 30.0, 15.5, 11.5)
    pts = np.array([[0, 0, 2, 1], [0, 0, 2, 1], [0.1, 0, 2, 1], [0, 0, -1, 1], [50, 0, 2, 1], [0, 0, 1.5, 1], [0, 0, 100, 1],
                    [0.0166667, 0.0166667, 1.0, 1]], np.float32)
    n = len(pts)
    z = np.zeros((n, 4), np.float32); om = np.zeros((n, 16), np.float32)
    oi, od = oracle.project(K, np.eye(4), 0.5, 5.0, rows, cols, pts)
    c = api.Cloud(ctx, n); c.upload(pts, z, np.zeros(n, np.float32), om, om)
    proj = api.PinholePointProjector(); proj.setCameraMatrix([[K[0], 0, K[2]], [0, K[1], K[3]], [0, 0, 1]])
    proj.setMinDistance(0.5); proj.setMaxDistance(5.0); proj.setImageSize(rows, cols)
    gi, gd = proj.project(c)
    assert np.array_equal(oi, gi) and np.array_equal(od.view(np.uint32), gd.view(np.uint32))
    assert gd.max() == np.finfo(np.float32).max and (gi >= 0).sum() >= 2


@pytest.fixture(scope="module")
def aligned_inputs(ctx, oracle):
    """Oracle clouds of the small pair uploaded to the GPU (isolates the aligner from converter ulps)."""
    out = {}
    for name, seed in (("small", 1), ("vga", 0)):
        ref, cur, Ttrue, _, _ = make_depth_pair(name, seed)
        cp, ap = oracle_params(oracle, name)
        oref, _, _ = oracle.convert(cp, ref)
        ocur, _, _ = oracle.convert(cp, cur)
        out[name] = dict(oref=oref, ocur=ocur, gref=upload(ctx, oref), gcur=upload(ctx, ocur), Ttrue=Ttrue)
    return out


@pytest.mark.parametrize("name", ["small", "vga"])
def test_correspondences_exact(ctx, oracle, aligned_inputs, name):
    d = aligned_inputs[name]
    rows, cols, K, conv, _ = case_params(name)
    _, ap = oracle_params(oracle, name)
    proj, _, aligner = gpu_objects(ctx, name)
    aligner.setReferenceCloud(d["gref"]); aligner.setCurrentCloud(d["gcur"])
    T = np.eye(4, dtype=np.float32)
    ri, _ = oracle.project(K, T, conv["min_distance"], conv["max_distance"], rows, cols, d["oref"].arrays()["points"])
    ci, _ = oracle.project(K, np.eye(4), conv["min_distance"], conv["max_distance"], rows, cols, d["ocur"].arrays()["points"])
    ocorr, oK = oracle.correspondences(ap, d["oref"], d["ocur"], ri, ci, T)
    gcorr, gK = aligner.computeCorrespondences(ri, ci, T)
    assert oK == gK and len(ocorr) > 100
    assert np.array_equal(ocorr, gcorr)
    full = aligner.correspondenceFinder().correspondences()
    assert (full[len(gcorr):] == -1).all()


@pytest.mark.parametrize("name", ["small", "vga"])
def test_linearize_matches_oracle(ctx, oracle, aligned_inputs, name):
    d = aligned_inputs[name]
    rows, cols, K, conv, _ = case_params(name)
    _, ap = oracle_params(oracle, name, accumulate_fp64=1)
    proj, _, aligner = gpu_objects(ctx, name)
    aligner.setReferenceCloud(d["gref"]); aligner.setCurrentCloud(d["gcur"])
    T = np.eye(4, dtype=np.float32)
    ri, _ = oracle.project(K, T, conv["min_distance"], conv["max_distance"], rows, cols, d["oref"].arrays()["points"])
    ci, _ = oracle.project(K, T, conv["min_distance"], conv["max_distance"], rows, cols, d["ocur"].arrays()["points"])
    corr, _ = oracle.correspondences(ap, d["oref"], d["ocur"], ri, ci, T)
    o = oracle.linearize(ap, d["oref"], d["ocur"], corr, T)
    g = aligner.linearize(corr, T)
    assert g["inliers"] == o["inliers"]
    assert abs(g["chi2"] - o["chi2_fp64"]) <= CHI2_RTOL * o["chi2_fp64"]
    hs = np.abs(o["H"]).max()
    assert np.abs(g["H"] - o["H"]).max() <= 1e-5 * hs
    assert np.abs(g["b"] - o["b"]).max() <= 1e-5 * np.abs(o["b"]).max() + 1e-6 * hs
    # empty list: zero system
    e = aligner.linearize(np.zeros((0, 2), np.int32), T)
    assert e["inliers"] == 0 and e["chi2"] == 0 and not e["H"].any()


def _check_alignment(o, g, flips=4):
    """Free-running trace (see the module docstring): chi2 within FREE_CHI2_RTOL_VGA when C is large (VGA and up),
    else within 1e-2; counters within a few flips; pose strict."""
    n = len(o["iterations"])
    assert g["iterations"] == n
    for i, it in enumerate(o["iterations"]):
        rel = abs(float(g["chi2"][i]) - it["chi2_fp64"]) / it["chi2_fp64"]
        tol = FREE_CHI2_RTOL_VGA if it["C"] >= 100000 else 1e-2
        assert rel <= tol, (i, rel, tol, float(g["chi2"][i]), it["chi2_fp64"])
        assert abs(int(g["C"][i]) - it["C"]) <= flips and abs(int(g["K"][i]) - it["K"]) <= 4 * flips
    assert np.abs(g["T"][:3, 3] - o["T"][:3, 3]).max() <= POSE_TTOL
    assert np.abs(g["T"][:3, :3] - o["T"][:3, :3]).max() <= POSE_RTOL


def _check_teacher_forced(aligner, o, strict_counts=True):
    """Re-run every iteration of the oracle trace `o` from the oracle's own iterate; returns the worst chi2 rel diff."""
    outer = aligner._outerIterations
    guess = aligner._initialGuess.copy()
    aligner.setOuterIterations(1)
    worst = 0.0
    try:
        for i, it in enumerate(o["iterations"]):
            aligner.setInitialGuess(it["T_before"])
            g = aligner.align()
            if strict_counts:
                assert (int(g["K"][0]), int(g["C"][0]), int(g["iter_inliers"][0])) == (it["K"], it["C"], it["inliers"]), i
            rel = abs(float(g["chi2"][0]) - it["chi2_fp64"]) / it["chi2_fp64"]
            worst = max(worst, rel)
            assert rel <= CHI2_RTOL, (i, rel)
    finally:
        aligner.setOuterIterations(outer); aligner.setInitialGuess(guess)
    return worst


@pytest.mark.parametrize("name", ["small", "vga"])
def test_per_iteration_chi2_teacher_forced(ctx, oracle, aligned_inputs, name):
    """Every iteration of the oracle's trace re-run on the GPU from the oracle's own iterate T_i:
    K_i, C_i, inliers_i exact; chi2_i within 1e-5 (strict, no flip allowance); H and b within 1e-5."""
    d = aligned_inputs[name]
    _, ap = oracle_params(oracle, name, accumulate_fp64=1)
    o = oracle.align(ap, d["oref"], d["ocur"])
    _, _, aligner = gpu_objects(ctx, name)
    aligner.setReferenceCloud(d["gref"]); aligner.setCurrentCloud(d["gcur"])
    worst = _check_teacher_forced(aligner, o)
    print(f"{name}: worst per-iteration chi2 rel diff {worst:.2e}")


@pytest.mark.parametrize("name", ["small", "vga"])
def test_align_from_identical_clouds(ctx, oracle, aligned_inputs, name):
    """Aligner::align on bit-identical input clouds: chi2 per iteration within 1e-5 of the fp64-accumulated
    oracle, counters K/C/inliers exact at iteration 0, final pose within 1e-5."""
    d = aligned_inputs[name]
    _, ap = oracle_params(oracle, name, accumulate_fp64=1)
    o = oracle.align(ap, d["oref"], d["ocur"], images=True)
    _, _, aligner = gpu_objects(ctx, name)
    aligner.setReferenceCloud(d["gref"]); aligner.setCurrentCloud(d["gcur"])
    g = aligner.align(images=True)
    it0 = o["iterations"][0]
    assert (g["K"][0], g["C"][0], g["iter_inliers"][0]) == (it0["K"], it0["C"], it0["inliers"])
    _check_alignment(o, g)
    assert g["error"] == g["chi2"][-1] and g["inliers"] == g["iter_inliers"][-1]
    f = aligner.correspondenceFinder()
    assert np.array_equal(f.currentIndexImage(), o["cur_index"])
    assert np.array_equal(f.currentDepthImage().view(np.uint32), o["cur_depth"].view(np.uint32))
    # last reference projection uses T_9, which carries ~1e-7 differences: allow a handful of pixel flips
    diff = int((f.referenceIndexImage() != o["ref_index"]).sum())
    assert diff <= max(4, o["ref_index"].size // 5000), diff
    # the pose must also be the true synthetic motion (sanity of the whole chain)
    assert np.abs(g["T"][:3, 3] - d["Ttrue"][:3, 3]).max() < 5e-3


def test_align_vs_reference_faithful_fp32_sums(ctx, oracle, aligned_inputs):
    """Against the oracle in reference-faithful mode (fp32 serial sums of H, b, chi2: linearizer.cpp:81-88).
    A 2e5-term fp32 serial sum is itself only ~3e-5 accurate, and the ~1e-4 perturbation it puts on the first
    Gauss-Newton steps moves chi2 of the early (far-from-converged) iterations by up to ~2e-3; both traces
    converge to the same pose.  Bars: iteration 0 (identical transform) 1e-4, any iteration 5e-3, pose 1e-4."""
    d = aligned_inputs["vga"]
    _, ap = oracle_params(oracle, "vga", accumulate_fp64=0)
    o = oracle.align(ap, d["oref"], d["ocur"])
    _, _, aligner = gpu_objects(ctx, "vga")
    aligner.setReferenceCloud(d["gref"]); aligner.setCurrentCloud(d["gcur"])
    g = aligner.align()
    rel = [abs(float(g["chi2"][i]) - it["chi2"]) / it["chi2"] for i, it in enumerate(o["iterations"])]
    print("chi2 rel diff vs fp32-serial oracle:", ["%.1e" % r for r in rel])
    assert rel[0] < 1e-4, rel
    assert max(rel) < 5e-3, rel
    assert np.abs(g["T"] - o["T"]).max() < 1e-4


@pytest.mark.parametrize("name,seed", [("small", 1), ("small", 2), ("vga", 0)])
def test_full_pipeline_depth_to_pose(ctx, oracle, name, seed):
    """convert + align entirely on the GPU vs entirely in the oracle."""
    from g2o_frontend_amd import api
    rows, cols, K, conv, _ = case_params(name)
    ref, cur, Ttrue, _, _ = make_depth_pair(name, seed)
    cp, ap = oracle_params(oracle, name, accumulate_fp64=1)
    oref, _, _ = oracle.convert(cp, ref); ocur, _, _ = oracle.convert(cp, cur)
    o = oracle.align(ap, oref, ocur)
    _, converter, aligner = gpu_objects(ctx, name)
    gref, gcur = api.Cloud(ctx, rows * cols), api.Cloud(ctx, rows * cols)
    converter.compute(gref, ref); converter.compute(gcur, cur)
    aligner.setReferenceCloud(gref); aligner.setCurrentCloud(gcur)
    g = aligner.align()
    _check_alignment(o, g)
    _check_teacher_forced(aligner, o)      # strict 1e-5 per iteration, from depth images to chi2, all on the GPU


def test_free_running_chi2_many_seeds(ctx, oracle):
    """The free-running chi2 distance is seed-dependent (round-2 review: the bench line's three pairs showed 1.4e-5 where seed 0 alone stays
    below 1e-5): 16 VGA pairs, depth images to final pose on the GPU against the fp64-accumulating oracle, no teacher forcing.  Reports the
    worst relative chi2 difference with the counters of that iteration on both sides (a difference above 1e-5 comes with a changed
    correspondence set: the same pairs re-run from the oracle's iterates agree to 1e-7 with exact counters) and enforces FREE_CHI2_RTOL_VGA;
    the strict 1e-5 bar of north_star is enforced teacher-forced on four of the seeds, the worst one included."""
    from g2o_frontend_amd import api
    name = "vga"
    rows, cols, K, conv, _ = case_params(name)
    cp, ap = oracle_params(oracle, name, accumulate_fp64=1)
    _, converter, aligner = gpu_objects(ctx, name)
    gref, gcur = api.Cloud(ctx, rows * cols), api.Cloud(ctx, rows * cols)
    per_seed, traces = [], {}
    for seed in range(100, 116):
        ref, cur, Ttrue, _, _ = make_depth_pair(name, seed)
        oref, _, _ = oracle.convert(cp, ref); ocur, _, _ = oracle.convert(cp, cur)
        o = oracle.align(ap, oref, ocur)
        converter.compute(gref, ref); converter.compute(gcur, cur)
        aligner.setReferenceCloud(gref); aligner.setCurrentCloud(gcur)
        g = aligner.align()
        rel = [abs(float(g["chi2"][i]) - it["chi2_fp64"]) / it["chi2_fp64"] for i, it in enumerate(o["iterations"])]
        w = int(np.argmax(rel)); itw = o["iterations"][w]
        per_seed.append(dict(seed=seed, worst=rel[w], it=w, dC=int(g["C"][w]) - itw["C"], dK=int(g["K"][w]) - itw["K"],
                             dInl=int(g["iter_inliers"][w]) - itw["inliers"], mean_term=1.0 / itw["C"],
                             dpose=float(np.abs(g["T"] - o["T"]).max())))
        traces[seed] = (ref, cur, o)
        assert g["iterations"] == len(o["iterations"]) and int(g["C"][0]) == o["iterations"][0]["C"]      # iteration 0: identical transform
        assert rel[0] <= CHI2_RTOL, (seed, rel[0])
        assert np.abs(g["T"][:3, 3] - o["T"][:3, 3]).max() <= POSE_TTOL and np.abs(g["T"][:3, :3] - o["T"][:3, :3]).max() <= POSE_RTOL, seed
    per_seed.sort(key=lambda r: -r["worst"])
    for r in per_seed:
        print("free-running seed %d: worst chi2 rel diff %.2e at iteration %d (dC %+d, dK %+d, dInliers %+d; one mean term = %.1e of chi2), pose diff %.1e"
              % (r["seed"], r["worst"], r["it"], r["dC"], r["dK"], r["dInl"], r["mean_term"], r["dpose"]))
    above = [r for r in per_seed if r["worst"] > CHI2_RTOL]
    print(f"free-running chi2 over 16 VGA seeds: worst {per_seed[0]['worst']:.2e}, {len(above)} seeds above 1e-5, bar {FREE_CHI2_RTOL_VGA:.0e}")
    assert per_seed[0]["worst"] <= FREE_CHI2_RTOL_VGA
    # the same pairs from the oracle's own iterates: 1e-5 strict, counters exact (worst free-running seeds first)
    for r in per_seed[:4]:
        ref, cur, o = traces[r["seed"]]
        converter.compute(gref, ref); converter.compute(gcur, cur)
        aligner.setReferenceCloud(gref); aligner.setCurrentCloud(gcur)
        worst = _check_teacher_forced(aligner, o)
        print(f"  seed {r['seed']} teacher-forced: worst chi2 rel diff {worst:.1e}, counters exact")


def test_full_pipeline_1280x960(oracle):
    """BASELINE configs[4] frame size: converter bit-exact, alignment trace within the bars, at 1280x960."""
    from g2o_frontend_amd import api
    name = "k2"
    rows, cols, K, conv, _ = case_params(name)
    ref, cur, Ttrue, _, _ = make_depth_pair(name, 0)
    cp, ap = oracle_params(oracle, name, accumulate_fp64=1)
    oref, oidx, _ = oracle.convert(cp, ref); ocur, _, _ = oracle.convert(cp, cur)
    o = oracle.align(ap, oref, ocur)
    big = api.Context(0, rows, cols, 2, omega_storage="exact9")
    _, converter, aligner = gpu_objects(big, name)
    gref, gcur = api.Cloud(big, rows * cols), api.Cloud(big, rows * cols)
    converter.compute(gref, ref)
    assert np.array_equal(oidx, converter.indexImage())
    _compare_clouds(oref.arrays(), gref.arrays(), name)
    converter.compute(gcur, cur)
    aligner.setReferenceCloud(gref); aligner.setCurrentCloud(gcur)
    g = aligner.align()
    _check_alignment(o, g)
    _check_teacher_forced(aligner, o)
    assert np.abs(g["T"][:3, 3] - Ttrue[:3, 3]).max() < 5e-3
    # the batch path at this size (it takes the converter's own index images instead of projecting the current cloud and, for the identity
    # guess, the reference cloud in the first iteration): bitwise the single alignment, in both pair orders
    res = aligner.alignBatch([gref, gcur], [gcur, gref])
    assert np.array_equal(res[0]["T"].view(np.uint32), g["T"].view(np.uint32)) and np.array_equal(res[0]["chi2"].view(np.uint32), g["chi2"].view(np.uint32))
    aligner.setReferenceCloud(gcur); aligner.setCurrentCloud(gref)
    h = aligner.align()
    assert np.array_equal(res[1]["T"].view(np.uint32), h["T"].view(np.uint32)) and np.array_equal(res[1]["chi2"].view(np.uint32), h["chi2"].view(np.uint32))
    big.close()


def test_align_with_sensor_offset_and_guess(ctx, oracle):
    from g2o_frontend_amd import api, synth
    name = "small"
    rows, cols, K, conv, _ = case_params(name)
    ref, cur, Ttrue, _, _ = make_depth_pair(name, 4)
    off = synth.v2t(np.array([0.02, -0.01, 0.03, 0.01, -0.02, 0.015])).astype(np.float32)
    guess = synth.v2t(np.array([0.01, 0.0, -0.01, 0.002, 0.001, -0.003])).astype(np.float32)
    cp, ap = oracle_params(oracle, name, sensor_offset=off, accumulate_fp64=1)
    ap = oracle.aligner_params(rows, cols, K=K, reference_sensor_offset=off, current_sensor_offset=off, initial_guess=guess,
                               accumulate_fp64=1, **case_params(name)[4])
    oref, _, _ = oracle.convert(cp, ref); ocur, _, _ = oracle.convert(cp, cur)
    o = oracle.align(ap, oref, ocur)
    _, converter, aligner = gpu_objects(ctx, name, sensor_offset=off)
    gref, gcur = api.Cloud(ctx, rows * cols), api.Cloud(ctx, rows * cols)
    converter.compute(gref, ref, sensorOffset=off); converter.compute(gcur, cur, sensorOffset=off)
    aligner.setReferenceCloud(gref); aligner.setCurrentCloud(gcur); aligner.setInitialGuess(guess)
    g = aligner.align()
    _check_alignment(o, g)


def test_distinct_information_thresholds(ctx, oracle):
    """Three different curvature thresholds in the converter (stats 0.2, point information 0.03, normal information 0.07) and a fourth in
    the finder (flat 0.02): the class of the normal information matrix is not stored with the cloud but derived from normal, curvature and the
    normal-information threshold the cloud was converted with (informationmatrixcalculator.cpp:38-58) -- which must not be mistaken for any of
    the other three.  Converter arrays bit for bit (the downloaded omega_n included), every iteration teacher-forced, batch == single."""
    from g2o_frontend_amd import api
    name = "small"
    rows, cols, K, conv, alig = case_params(name)
    conv = dict(conv, stats_curvature_threshold=0.2, point_info_curvature_threshold=0.03, normal_info_curvature_threshold=0.07)
    ref, cur, _, ref_mm, cur_mm = make_depth_pair(name, 6)
    cp = oracle.converter_params(K=K, **conv)
    ap = oracle.aligner_params(rows, cols, K=K, accumulate_fp64=1, **alig)
    oref, _, _ = oracle.convert(cp, ref); ocur, _, _ = oracle.convert(cp, cur)
    oa = ocur.arrays()
    curv = oa["curvature"]; has_n = np.abs(oa["normals"][:, :3]).sum(1) > 0
    assert (has_n & (curv >= 0.03) & (curv < 0.07)).sum() > 50 and (has_n & (curv >= 0.07)).sum() > 50, "degenerate input: no point between the thresholds"
    _, converter, aligner = gpu_objects(ctx, name)
    converter._stats.setCurvatureThreshold(0.2); converter._pinfo.setCurvatureThreshold(0.03); converter._ninfo.setCurvatureThreshold(0.07)
    gref, gcur = api.Cloud(ctx, rows * cols), api.Cloud(ctx, rows * cols)
    converter.compute(gref, ref); converter.compute(gcur, cur)
    _compare_clouds(oref.arrays(), gref.arrays(), name); _compare_clouds(oa, gcur.arrays(), name)
    o = oracle.align(ap, oref, ocur)
    aligner.setReferenceCloud(gref); aligner.setCurrentCloud(gcur)
    g = aligner.align()
    _check_alignment(o, g)
    _check_teacher_forced(aligner, o)
    # the batch path (raw uint16 frames, lean hand-over inside the converter) gives the same clouds and the same alignment
    bref, bcur = api.Cloud(ctx, rows * cols), api.Cloud(ctx, rows * cols)
    converter.computeBatch([bref, bcur], [ref_mm, cur_mm], raw_scale=0.001)
    _compare_clouds(oa, bcur.arrays(), name)
    b = aligner.alignBatch([bref, gref], [bcur, gcur])
    for r in b:
        assert np.array_equal(r["T"], g["T"]) and np.array_equal(r["chi2"], g["chi2"])


def test_inner_iterations_and_nonrobust(ctx, oracle, aligned_inputs):
    d = aligned_inputs["small"]
    _, ap = oracle_params(oracle, "small", accumulate_fp64=1, inner_iterations=2, outer_iterations=4, robust_kernel=0, inlier_max_chi2=50.0)
    o = oracle.align(ap, d["oref"], d["ocur"])
    _, _, aligner = gpu_objects(ctx, "small")
    aligner.setInnerIterations(2); aligner.setOuterIterations(4)
    aligner.linearizer().setRobustKernel(False); aligner.linearizer().setInlierMaxChi2(50.0)
    aligner.setReferenceCloud(d["gref"]); aligner.setCurrentCloud(d["gcur"])
    g = aligner.align()
    assert g["iterations"] == 8
    assert g["iter_inliers"][0] == o["iterations"][0]["inliers"] < o["iterations"][0]["C"]
    _check_alignment(o, g)


def test_batch_equals_single_and_is_deterministic(ctx, oracle):
    """Batched convert/align give bit-identical results to one-at-a-time calls, run to run."""
    from g2o_frontend_amd import api
    name = "small"
    rows, cols, K, conv, _ = case_params(name)
    _, converter, aligner = gpu_objects(ctx, name)
    pairs = [make_depth_pair(name, s) for s in (5, 6, 7)]
    single = []
    for ref, cur, _, _, _ in pairs:
        a, b = api.Cloud(ctx, rows * cols), api.Cloud(ctx, rows * cols)
        converter.compute(a, ref); converter.compute(b, cur)
        aligner.setReferenceCloud(a); aligner.setCurrentCloud(b)
        single.append(aligner.align())
    for sub in (1, 2, 4):
        ctx.set_subbatch(sub, sub)
        refs = [api.Cloud(ctx, rows * cols) for _ in pairs]; curs = [api.Cloud(ctx, rows * cols) for _ in pairs]
        converter.computeBatch(refs + curs, [p[0] for p in pairs] + [p[1] for p in pairs])
        res = aligner.alignBatch(refs, curs)
        for s, r in zip(single, res):
            assert np.array_equal(s["T"].view(np.uint32), r["T"].view(np.uint32))
            assert np.array_equal(s["chi2"].view(np.uint32), r["chi2"].view(np.uint32))
            assert np.array_equal(s["C"], r["C"]) and np.array_equal(s["K"], r["K"])
    ctx.set_subbatch(64, 64)


def test_batch_from_raw_u16(ctx, oracle):
    from g2o_frontend_amd import api
    name = "small"
    rows, cols, K, conv, _ = case_params(name)
    _, converter, _ = gpu_objects(ctx, name)
    ref, cur, _, ref_mm, cur_mm = make_depth_pair(name, 8)
    a, b = api.Cloud(ctx, rows * cols), api.Cloud(ctx, rows * cols)
    converter.computeBatch([a, b], [ref_mm, cur_mm], raw_scale=0.001)
    c = api.Cloud(ctx, rows * cols)
    converter.compute(c, ref)
    x, y = a.arrays(), c.arrays()
    for k in x:
        assert np.array_equal(x[k].view(np.uint32), y[k].view(np.uint32)), k


def test_error_paths(ctx):
    from g2o_frontend_amd import api
    from g2o_frontend_amd._lib import PwnHipError
    _, converter, _ = gpu_objects(ctx, "small")
    big = np.ones((1000, 1000), np.float32)
    with pytest.raises(PwnHipError) as e:
        converter.compute(api.Cloud(ctx, 10), big)
    assert e.value.code == 6
    small_cloud = api.Cloud(ctx, 10)
    with pytest.raises(PwnHipError):
        converter.compute(small_cloud, np.full((120, 160), 1.0, np.float32))
    # a wide, short image with rows*cols <= the context's pixel budget but cols > max(max_rows, max_cols): the strip hand-over
    # workspaces are sized for sides <= that maximum, so it is refused, alone and in a batch
    assert ctx.max_rows * ctx.max_cols >= 17 * 18070 and 18070 > max(ctx.max_rows, ctx.max_cols)
    wide = np.full((17, 18070), 1.0, np.float32)
    with pytest.raises(PwnHipError) as e:
        converter.compute(api.Cloud(ctx, 17 * 18070), wide)
    assert e.value.code == 6
    with pytest.raises(PwnHipError) as e:
        converter.computeBatch([api.Cloud(ctx, 17 * 18070) for _ in range(16)], [wide] * 16)
    assert e.value.code == 6


@pytest.mark.parametrize("rows,cols", [(113, 150), (97, 131), (33, 70), (17, 65), (16, 64), (121, 63), (5, 300), (203, 170)])
def test_odd_image_sizes(oracle, rows, cols):
    """Image sizes that are no multiple of anything the kernels tile by (64-column strips, 16-row bands, 256-pixel blocks, 2048-pixel
    tiles of the fused kernel): converter arrays and images bit for bit -- alone (three-kernel integral image) and in a 26-frame batch
    (single-pass strip kernel) --, first-iteration counters exactly, chi2 from the same iterate to 1e-5, batch == single bitwise."""
    from g2o_frontend_amd import api, synth
    scale = 4
    K = synth.scaled_K(synth.K_VGA, scale)
    K = (K[0], K[1], (cols - 1) / 2.0, (rows - 1) / 2.0)
    conv, alig = dict(oracle.QVGA4_CONF_CONVERTER), dict(oracle.QVGA4_CONF_ALIGNER)
    ctx = api.Context(0, rows, cols, 32, omega_storage="exact9")
    try:
        proj, converter, aligner = gpu_objects(ctx, "small")
        for p in (proj, aligner.projector()):
            p.setCameraMatrix([[K[0], 0, K[2]], [0, K[1], K[3]], [0, 0, 1]]); p.setImageSize(rows, cols)
        aligner.correspondenceFinder().setImageSize(rows, cols)
        ref_mm, cur_mm, _ = synth.make_pair(11, rows, cols, K)
        ref, cur = oracle.convert_16u_to_32f(ref_mm), oracle.convert_16u_to_32f(cur_mm)
        cp = oracle.converter_params(K=K, **conv)
        oref, oidx, oitv = oracle.convert(cp, ref); ocur, _, _ = oracle.convert(cp, cur)
        gref, gcur = api.Cloud(ctx, rows * cols), api.Cloud(ctx, rows * cols)
        converter.compute(gref, ref)
        assert np.array_equal(converter.indexImage(), oidx) and np.array_equal(converter.intervalImage(), oitv)
        converter.compute(gcur, cur)
        def same(o, g):
            assert len(o["points"]) == len(g["points"])
            for k in ("points", "normals", "curvature", "omega_p", "omega_n"):
                a, b = o[k].reshape(len(o[k]), -1), g[k].reshape(len(g[k]), -1)
                assert (((a.view(np.uint32) == b.view(np.uint32)) | ((a == 0) & (b == 0)))).all(), k
        same(oref.arrays(), gref.arrays()); same(ocur.arrays(), gcur.arrays())
        # the batch path (>= 16 frames: single-pass strip kernel)
        many = [api.Cloud(ctx, rows * cols) for _ in range(26)]
        converter.computeBatch(many, [ref_mm, cur_mm] * 13, raw_scale=0.001)
        for k in (0, 1, 24, 25):
            same((oref if k % 2 == 0 else ocur).arrays(), many[k].arrays())
        if len(oref) == 0 or len(ocur) == 0:
            return
        ap = oracle.aligner_params(rows, cols, K=K, accumulate_fp64=1, **alig)
        o = oracle.align(ap, oref, ocur)
        aligner.setReferenceCloud(gref); aligner.setCurrentCloud(gcur)
        g = aligner.align()
        it0 = o["iterations"][0]
        assert (int(g["K"][0]), int(g["C"][0]), int(g["iter_inliers"][0])) == (it0["K"], it0["C"], it0["inliers"])
        if it0["chi2_fp64"] > 0:
            assert abs(float(g["chi2"][0]) - it0["chi2_fp64"]) <= 1e-5 * it0["chi2_fp64"]
        res = aligner.alignBatch([many[0], many[2], gref], [many[1], many[3], gcur])
        for r in res:
            assert np.array_equal(r["T"].view(np.uint32), g["T"].view(np.uint32)) and np.array_equal(r["chi2"].view(np.uint32), g["chi2"].view(np.uint32))
    finally:
        ctx.close()


@pytest.mark.parametrize("shrink", [1, 4])
def test_aligner_finder_images_bit_exact_and_collisions(oracle, shrink):
    """The aligner's own z-buffer (32-bit tag | index words; points of one projection that meet in a pixel are settled by depth with a
    compare-and-swap loop) against the oracle's sequential projector: CorrespondenceFinder::{reference,current}{Index,Depth}Image() after
    Aligner::align, bit for bit.  shrink = 4 projects VGA clouds into a 120x160 image: 16 points per pixel fight for every word (the
    reference's `>` keeps the nearest, ties the lowest index: pinholepointprojector.cpp:61).  Single alignment (one point per thread)
    and a batch of 9 (four points per thread) must agree bit for bit as well."""
    from g2o_frontend_amd import api, synth
    rows, cols, K, conv, alig = case_params("vga")
    ref, cur, _, _, _ = make_depth_pair("vga", 3)
    ctx = api.Context(0, rows, cols, 16, omega_storage="exact9")
    try:
        _, converter, aligner = gpu_objects(ctx, "vga")
        gref, gcur = api.Cloud(ctx, rows * cols), api.Cloud(ctx, rows * cols)
        converter.compute(gref, ref); converter.compute(gcur, cur)
        cp, _ = oracle_params(oracle, "vga")
        oref, _, _ = oracle.convert(cp, ref); ocur, _, _ = oracle.convert(cp, cur)
        r2, c2 = rows // shrink, cols // shrink
        K2 = synth.scaled_K(K, shrink) if shrink > 1 else K
        for p in (aligner.projector(),):
            p.setCameraMatrix([[K2[0], 0, K2[2]], [0, K2[1], K2[3]], [0, 0, 1]]); p.setImageSize(r2, c2)
        aligner.correspondenceFinder().setImageSize(r2, c2)
        aligner.setReferenceCloud(gref); aligner.setCurrentCloud(gcur)
        guesses = [np.eye(4, dtype=np.float32), synth.v2t(np.array([0.03, -0.02, 0.01, 0.01, -0.02, 0.015])).astype(np.float32)]
        for outer in (1, 3):
            aligner.setOuterIterations(outer)
            for g0 in guesses:
                aligner.setInitialGuess(g0)
                ap = oracle.aligner_params(r2, c2, K=K2, initial_guess=g0, accumulate_fp64=1, **dict(alig, outer_iterations=outer))
                o = oracle.align(ap, oref, ocur, images=True)
                if outer > 1:
                    # free-running iterates may differ in their last bits: compare the images of the LAST projection from the oracle's iterate
                    aligner.setOuterIterations(1); aligner.setInitialGuess(o["iterations"][-1]["T_before"])
                g = aligner.align(images=True)
                f = aligner.correspondenceFinder()
                assert np.array_equal(f.referenceIndexImage(), o["ref_index"]) and np.array_equal(f.currentIndexImage(), o["cur_index"])
                assert np.array_equal(f.referenceDepthImage().view(np.uint32), o["ref_depth"].view(np.uint32))
                assert np.array_equal(f.currentDepthImage().view(np.uint32), o["cur_depth"].view(np.uint32))
                if shrink > 1:
                    assert int((o["ref_index"] >= 0).sum()) < len(oref) // 8              # the collisions are real: most points lose
                aligner.setOuterIterations(outer)
        # batch (>= 8 pairs: four points per thread, all atomics in flight) == single (one point per thread)
        aligner.setOuterIterations(3); aligner.setInitialGuess(guesses[1])
        single = aligner.align()
        batch = aligner.alignBatch([gref] * 9, [gcur] * 9, initialGuesses=[guesses[1]] * 9)
        for b in batch:
            assert np.array_equal(b["T"].view(np.uint32), single["T"].view(np.uint32)) and np.array_equal(b["chi2"].view(np.uint32), single["chi2"].view(np.uint32))
            assert np.array_equal(b["K"], single["K"]) and np.array_equal(b["C"], single["C"])
    finally:
        ctx.close()


def test_gpu_linearizer_against_the_references_octave_model(ctx):
    """The GPU linearizer (pwn_hip_linearize: k_linearize_list + the fixed-order reduction) directly against the one executable model of this
    path the reference repository holds -- octave/pwn/pwn_iteration.m restated in numpy (tests/test_reference_octave_model.py), no oracle in
    between: PWNTest.m's scenario (100 points with normals in a 100 m cube, 120-degree ground-truth rotation), chi2 at every iterate of the
    model's own 40-iteration trajectory; and H, b at the identity on a small-motion pair (where the model's Jacobian is the C++ one)."""
    from g2o_frontend_amd import api
    from test_reference_octave_model import m_v2t, m_remap, m_iteration_sums
    _, _, aligner = gpu_objects(ctx, "small")
    aligner.linearizer().setInlierMaxChi2(1e30)

    def cloud(P6, Omega):
        n = P6.shape[1]
        P = np.ones((n, 4), np.float32); P[:, :3] = P6[:3].T
        N = np.zeros((n, 4), np.float32); N[:, :3] = P6[3:].T
        op = np.zeros((n, 4, 4), np.float32); op[:, :3, :3] = Omega[:3, :3]
        on = np.zeros((n, 4, 4), np.float32); on[:, :3, :3] = Omega[3:, 3:]
        c = api.Cloud(ctx, n)
        c.upload(P, N, np.full(n, 0.01, np.float32), op.transpose(0, 2, 1).reshape(n, 16), on.transpose(0, 2, 1).reshape(n, 16))
        return c
    rng = np.random.default_rng(11)
    n = 100
    Pi = rng.uniform(-0.5, 0.5, (6, n)); Pi[:3] *= 100.0; Pi[3:] /= np.linalg.norm(Pi[3:], axis=0, keepdims=True)
    gtX = m_v2t(np.array([100.0, 200.0, 300.0, 0.5, 0.5, 0.5]))
    Pj = np.stack([m_remap(gtX, Pi[:, i]) for i in range(n)], 1)
    Omega = np.eye(6); Omega[3:, 3:] *= 100.0
    aligner.setCurrentCloud(cloud(Pi, Omega)); aligner.setReferenceCloud(cloud(Pj, Omega))
    corr = np.stack([np.arange(n), np.arange(n)], 1).astype(np.int32)
    X = np.eye(4); worst = 0.0
    for it in range(40):
        H, b, err = m_iteration_sums(Pi, Pj, Omega, X)
        g = aligner.linearize(corr, X.astype(np.float32))
        assert g["inliers"] == n
        # fp32 terms of 100 m coordinates against float64: relative 2e-4; once the noise-free scenario has converged chi2 is rounding noise of
        # ~300 m coordinates (1e-5), hence the absolute term
        assert abs(g["chi2"] - err) <= 2e-4 * err + 1e-3, (it, g["chi2"], err)
        if err > 1.0:
            worst = max(worst, abs(g["chi2"] - err) / err)
        X = X @ m_v2t(-np.linalg.solve(H, b))
    # H, b at the identity, small motion
    X = m_v2t(np.array([0.004, -0.003, 0.002, 0.001, -0.002, 0.0015]))
    Pi2 = rng.uniform(-1, 1, (6, 400)); Pi2[:3] += np.array([[0.0], [0.0], [2.5]]); Pi2[3:] /= np.linalg.norm(Pi2[3:], axis=0, keepdims=True)
    Pj2 = np.stack([m_remap(np.linalg.inv(X), Pi2[:, i]) for i in range(400)], 1)
    Q = np.linalg.qr(rng.normal(size=(3, 3)))[0]
    Om2 = np.zeros((6, 6)); Om2[:3, :3] = Q @ np.diag([1000.0, 1.0, 1.0]) @ Q.T; Om2[3:, 3:] = np.eye(3) * 100.0
    aligner.setCurrentCloud(cloud(Pi2, Om2)); aligner.setReferenceCloud(cloud(Pj2, Om2))
    corr2 = np.stack([np.arange(400), np.arange(400)], 1).astype(np.int32)
    H, b, err = m_iteration_sums(Pi2, Pj2, Om2, np.eye(4))
    g = aligner.linearize(corr2, np.eye(4, dtype=np.float32))
    assert abs(g["chi2"] - err) <= 5e-5 * err
    assert np.abs(g["H"] - H).max() <= 5e-5 * np.abs(H).max() and np.abs(g["b"] - b).max() <= 5e-5 * np.abs(b).max() + 1e-6 * np.abs(H).max()
    print(f"GPU linearizer vs pwn_iteration.m model: worst chi2 rel diff {worst:.1e} over 40 iterates of PWNTest.m's trajectory")


def test_gpu_converter_against_float64_brute_force():
    """The GPU converter without the oracle in between: a float64 numpy brute force of the reference's per-pixel statistics
    (statscalculatorintegralimage.cpp:33-80: window x in (c-rad-1, c+rad-1], y in (r-rad-1, r+rad-1] after the clamps of
    pointintegralimage.cpp:57-60, mean / covariance of pointaccumulator.h:66-86, smallest eigenvector, curvature of stats.h:98-103, normal
    turned towards the sensor) on a 120x160 frame: window counts exactly, normals and curvatures to the accuracy of fp32 integral-image sums."""
    from g2o_frontend_amd import api, synth
    name = "small"
    rows, cols, K, conv, _ = case_params(name)
    depth_mm = synth.render_depth_mm(21, np.eye(4), rows, cols, K)
    depth = depth_mm.astype(np.float32) * np.float32(0.001); depth[depth_mm == 0] = 0
    c = api.Context(0, rows, cols, 2, omega_storage="exact9")
    _, converter, _ = gpu_objects(c, name)
    g = api.Cloud(c, rows * cols)
    converter.compute(g, depth, keep_stats=True)
    idx, itv = converter.indexImage(), converter.intervalImage()
    a = g.arrays(stats=True)
    fx, fy, cx, cy = K
    valid = (depth >= conv["min_distance"]) & (depth <= conv["max_distance"])
    assert np.array_equal(valid, idx >= 0) and np.array_equal(idx[valid], np.arange(valid.sum()))          # ordered compaction
    d64 = depth.astype(np.float64)
    cc, rr = np.meshgrid(np.arange(cols, dtype=np.float64), np.arange(rows, dtype=np.float64))
    P = np.stack([(cc - cx) / fx * d64, (rr - cy) / fy * d64, d64], -1)                                      # iK * (c d, r d, d)
    assert np.abs(a["points"][:, :3] - P[valid]).max() < 2e-6 * 4.5
    # interval image: int(max(fx R / d, fy R / d)) in fp32 (pinholepointprojector.h:264-274)
    R = np.float32(conv["world_radius"])
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = np.float32(1.0) / np.where(valid, depth, np.float32(1.0))
        want_itv = np.where(valid, np.maximum(np.float32(fx) * R * inv, np.float32(fy) * R * inv).astype(np.int32), -1)
    assert (want_itv != itv).sum() <= 2                                                                     # (K R) / d vs (K R) * (1/d): a truncation boundary at most
    rng = np.random.default_rng(3)
    checked = big = 0
    worst_ang = worst_curv = 0.0
    angs = []; ncomp = 0
    Pv = np.where(valid[..., None], P, 0.0)
    for _ in range(600):
        r, cpx = int(rng.integers(0, rows)), int(rng.integers(0, cols))
        if not valid[r, cpx]:
            continue
        rad = int(np.clip(itv[r, cpx], conv["min_image_radius"], conv["max_image_radius"]))
        cl = lambda x, hi: min(max(x, 0), hi)
        x0, x1 = cl(cpx - rad - 1, cols - 1), cl(cpx + rad - 1, cols - 1)
        y0, y1 = cl(r - rad - 1, rows - 1), cl(r + rad - 1, rows - 1)
        m = valid[y0 + 1:y1 + 1, x0 + 1:x1 + 1]
        n = int(m.sum())
        i = idx[r, cpx]
        assert a["npoints"][i] == (n if n >= conv["min_points"] else 0), (r, cpx, rad, n, a["npoints"][i])
        if n < conv["min_points"]:
            assert not a["normals"][i, :3].any()
            continue
        pts = Pv[y0 + 1:y1 + 1, x0 + 1:x1 + 1][m]
        mean = pts.mean(0)
        cov = pts.T @ pts / n - np.outer(mean, mean)
        w, V = np.linalg.eigh(cov)
        curv = max(w[0], 0.0) / (max(w[0], 0.0) + w[1] + w[2] + 1e-9)
        checked += 1
        # fp32 sums of ~n squared coordinates: the covariance is a difference of terms ~|p|^2 with relative error ~1e-6, i.e. absolute ~1e-5 m^2
        # against eigenvalues down to ~1e-7 on the planes: compare where the plane is well conditioned
        if w[1] < 50 * 2e-5:
            continue
        big += 1
        nrm = a["normals"][i, :3].astype(np.float64)
        p2 = float(P[r, cpx] @ P[r, cpx])
        tolc = 8e-5 * p2 / (max(w[0], 0.0) + w[1] + w[2])          # the same covariance error seen by the curvature ratio
        if tolc > 0.05:
            continue                                               # small window far away: fp32 sums leave no comparable curvature (the reference's do not either)
        assert abs(float(a["curvature"][i]) - curv) <= tolc + 1e-4 or not nrm.any(), (r, cpx, float(a["curvature"][i]), curv, tolc)
        if curv + tolc < conv["stats_curvature_threshold"]:
            assert nrm.any(), (r, cpx, curv, tolc)
            want = V[:, 0] * (-1.0 if V[:, 0] @ P[r, cpx] > 0 else 1.0)
            ang = np.arccos(np.clip(abs(nrm @ want), -1, 1))
            # the fp32 sums put an absolute error of a few 1e-5 |p|^2 on the covariance (the reference's integral image has the same); the normal
            # turns by that over the eigen-gap
            tol = min(0.2, 4e-5 * float(P[r, cpx] @ P[r, cpx]) / (w[1] - max(w[0], 0.0)))
            assert nrm @ P[r, cpx] <= 0 and ang < tol, (r, cpx, ang, tol, w)
            worst_ang = max(worst_ang, ang); angs.append(ang)
            worst_curv = max(worst_curv, abs(float(a["curvature"][i]) - curv))
            ncomp += 1
    assert checked > 200 and ncomp > 60 and worst_curv < 0.05 and np.median(angs) < 0.01
    print(f"GPU converter vs float64 brute force: {checked} windows counted exactly, {ncomp} normals compared, median angle {np.median(angs):.1e} rad, "
          f"worst {worst_ang:.1e} rad, worst |curvature diff| {worst_curv:.1e}")
    c.close()


def test_gpu_projector_against_a_numpy_zbuffer():
    """PinholePointProjector::project on the GPU (pwn_hip_project and the aligner's 32-bit z-buffer path through align(images=True)) against a
    numpy restatement of pinholepointprojector.cpp:33-66 / .h:224-233 written with the same fp32 operations -- no oracle in between: for every
    point KRt p with left-to-right fp32 sums, 1/d, roundf, bounds; per pixel the nearest point, ties to the lowest index: index and depth
    images bit for bit, on a real converted cloud projected from a moved pose (collisions, out-of-image points, range limits)."""
    from g2o_frontend_amd import api, synth
    name = "small"
    rows, cols, K, conv, _ = case_params(name)
    c = api.Context(0, rows, cols, 2, omega_storage="exact9")
    proj, converter, _ = gpu_objects(c, name)
    depth_mm = synth.render_depth_mm(23, np.eye(4), rows, cols, K)
    g = api.Cloud(c, rows * cols)
    converter.compute(g, c.DepthImage_convert_16UC1_to_32FC1(depth_mm))
    P = g.arrays()["points"][:, :3].astype(np.float32)
    f32 = np.float32
    for v in ([0.0] * 6, [0.03, -0.02, 0.05, 0.01, -0.015, 0.02], [-0.2, 0.1, 0.3, -0.05, 0.04, 0.03]):
        T = synth.v2t(np.array(v)).astype(np.float32)
        proj.setImageSize(rows, cols); proj.setTransform(T)
        gi, gd = proj.project(g)
        KRt = proj.matrices()[0].astype(np.float32)              # pwn_hip_projector_matrices: K * inverse(T), row-major view
        x, y, z = P[:, 0], P[:, 1], P[:, 2]
        def row(r):                                              # ((a x + b y) + c z) + d * 1, every operation rounded to fp32
            return ((KRt[r, 0] * x + KRt[r, 1] * y) + KRt[r, 2] * z) + KRt[r, 3] * f32(1.0)
        ix, iy, d = row(0), row(1), row(2)
        ok = ~((d < f32(conv["min_distance"])) | (d > f32(conv["max_distance"])))
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = f32(1.0) / d
            fxp, fyp = ix * inv, iy * inv
        def roundf(a):                                           # half away from zero, exactly
            t = np.trunc(a); fr = a - t
            return np.where(fr >= f32(0.5), t + 1, np.where(fr <= f32(-0.5), t - 1, t)).astype(np.float32)
        with np.errstate(invalid="ignore"):
            rx, ry = roundf(fxp), roundf(fyp)
            ok &= (rx >= 0) & (rx < cols) & (ry >= 0) & (ry < rows)
        idx = np.nonzero(ok)[0]
        pix = ry[idx].astype(np.int64) * cols + rx[idx].astype(np.int64)
        order = np.lexsort((idx, d[idx], pix))                   # per pixel: nearest first, ties to the lowest index
        pix_s, idx_s = pix[order], idx[order]
        first = np.ones(len(pix_s), bool); first[1:] = pix_s[1:] != pix_s[:-1]
        wi = np.full(rows * cols, -1, np.int32); wd = np.full(rows * cols, np.finfo(np.float32).max, np.float32)
        wi[pix_s[first]] = idx_s[first]; wd[pix_s[first]] = d[idx_s[first]]
        assert np.array_equal(gi.reshape(-1), wi), (v, int((gi.reshape(-1) != wi).sum()))
        assert np.array_equal(gd.reshape(-1).view(np.uint32), wd.view(np.uint32)), v
        collisions = int(len(pix_s) - first.sum())
        print(f"projector vs numpy z-buffer, motion {v}: {int(first.sum())} pixels, {collisions} collisions settled, bit-exact")
    c.close()


def test_gpu_correspondence_finder_against_numpy():
    """CorrespondenceFinder::compute on the GPU against a numpy restatement of correspondencefinder.cpp:45-106 with the same fp32 operations
    (T applied as the homogeneous 4x4 product with left-to-right sums; the curvature ratio in double, rounded to float) -- no oracle in
    between: the correspondence list (row-major order) and the candidate count exactly, for a near pose and for one that rejects many."""
    from g2o_frontend_amd import api, synth
    name = "small"
    rows, cols, K, conv, alig = case_params(name)
    c = api.Context(0, rows, cols, 2, omega_storage="exact9")
    proj, converter, aligner = gpu_objects(c, name)
    ref_mm, cur_mm, _ = synth.make_pair(31, rows, cols, K)
    gr, gc = api.Cloud(c, rows * cols), api.Cloud(c, rows * cols)
    converter.compute(gr, c.DepthImage_convert_16UC1_to_32FC1(ref_mm)); ri = converter.indexImage().copy()
    converter.compute(gc, c.DepthImage_convert_16UC1_to_32FC1(cur_mm)); ci = converter.indexImage().copy()
    aligner.setReferenceCloud(gr); aligner.setCurrentCloud(gc)
    A, B = gr.arrays(), gc.arrays()
    f32 = np.float32
    for v in ([0.0] * 6, [0.01, -0.02, 0.03, 0.02, -0.01, 0.015], [0.3, 0.0, 0.0, 0.0, 0.05, 0.0]):
        T = synth.v2t(np.array(v)).astype(np.float32); T[3] = (0, 0, 0, 1)
        gcorr, gK = aligner.computeCorrespondences(ri, ci, T)
        r_i, c_i = ri.reshape(-1), ci.reshape(-1)
        cand = (r_i >= 0) & (c_i >= 0)                                                                       # :60
        K_want = int(cand.sum())
        rI, cI = r_i[cand], c_i[cand]
        rP, rN, cP, cN = A["points"][rI], A["normals"][rI], B["points"][cI], B["normals"][cI]
        def sq3(a): return (a[:, 0] * a[:, 0] + a[:, 1] * a[:, 1]) + a[:, 2] * a[:, 2]
        ok = (sq3(cN) != 0) & (sq3(rN) != 0)                                                                # :69
        def iso(Tm, p, w):                                                                                  # Isometry3f * Vector4f, left to right
            return np.stack([((Tm[k, 0] * p[:, 0] + Tm[k, 1] * p[:, 1]) + Tm[k, 2] * p[:, 2]) + Tm[k, 3] * f32(w) for k in range(3)], 1)
        rp, rn = iso(T, rP, 1.0), iso(T, rN, 0.0)
        ok &= ~(((cN[:, 0] * rn[:, 0] + cN[:, 1] * rn[:, 1]) + cN[:, 2] * rn[:, 2]) < f32(alig["inlier_normal_angular_threshold"]))      # :78
        dd = cP[:, :3] - rp
        ok &= ~(sq3(dd) > f32(alig["inlier_distance_threshold"]) * f32(alig["inlier_distance_threshold"]))                                 # :84
        flat = f32(alig["flat_curvature_threshold"])
        rc = np.maximum(A["curvature"][rI], flat); cc = np.maximum(B["curvature"][cI], flat)                  # :89-93
        ratio = ((rc.astype(np.float64) + 1e-5) / (cc.astype(np.float64) + 1e-5)).astype(np.float32)          # :96
        mx = f32(alig["inlier_curvature_ratio_threshold"]); mn = f32(1.0) / mx
        ok &= ~((ratio < mn) | (ratio > mx))                                                                # :97-99
        want = np.stack([rI[ok], cI[ok]], 1).astype(np.int32)
        assert gK == K_want and np.array_equal(gcorr, want), (v, gK, K_want, len(gcorr), len(want))
        print(f"finder vs numpy, motion {v}: K {gK}, C {len(gcorr)}: exact")
    c.close()


def test_gpu_unproject_and_integral_image_against_numpy_fp32():
    """unProject (pinholepointprojector.cpp:93-133, .h:246-251) and PointIntegralImage::compute (pointintegralimage.cpp:7-44) on the GPU against
    numpy float32 arithmetic in the reference's order -- no oracle in between: points iKRt * (c d, r d, d, 1) with left-to-right fp32 sums, the ten
    accumulator planes (x y z 1 xx xy xz yy yz zz), a sequential fp32 prefix sum along image x, then along image y (numpy's float32 cumsum adds
    strictly left to right).  Bit for bit, VGA."""
    from g2o_frontend_amd import api, synth
    name = "vga"
    rows, cols, K, conv, _ = case_params(name)
    c = api.Context(0, rows, cols, 2, omega_storage="exact9")
    proj, _, _ = gpu_objects(c, name)
    depth = c.DepthImage_convert_16UC1_to_32FC1(synth.render_depth_mm(33, np.eye(4), rows, cols, K))
    cloud = api.Cloud(c, rows * cols)
    gidx = proj.unProject(cloud, depth)
    gI = api.StatsCalculatorIntegralImage.integralImage(cloud, gidx)
    f32 = np.float32
    proj.setTransform(np.eye(4, dtype=np.float32))
    iKRt = proj.matrices()[1].astype(np.float32)
    valid = ~((depth < f32(conv["min_distance"])) | (depth > f32(conv["max_distance"])))
    assert np.array_equal(gidx >= 0, valid) and np.array_equal(gidx[valid], np.arange(valid.sum()))
    cc, rr = np.meshgrid(np.arange(cols, dtype=np.float32), np.arange(rows, dtype=np.float32))
    a, b, d = cc * depth, rr * depth, depth
    def row(k): return ((iKRt[k, 0] * a + iKRt[k, 1] * b) + iKRt[k, 2] * d) + iKRt[k, 3] * f32(1.0)
    x, y, z = [np.where(valid, row(k), f32(0)).astype(np.float32) for k in range(3)]
    pts = cloud.arrays()["points"]
    assert np.array_equal(pts[:, 0].view(np.uint32), x[valid].view(np.uint32)) and np.array_equal(pts[:, 1].view(np.uint32), y[valid].view(np.uint32))
    assert np.array_equal(pts[:, 2].view(np.uint32), z[valid].view(np.uint32))
    one = valid.astype(np.float32)
    planes = [x, y, z, one, x * x, x * y, x * z, y * y, y * z, z * z]
    for k, p in enumerate(planes):
        want = np.cumsum(np.cumsum(p.astype(np.float32), axis=1, dtype=np.float32), axis=0, dtype=np.float32)
        assert np.array_equal(np.asarray(gI[k]).reshape(rows, cols).view(np.uint32), want.view(np.uint32)), k
    print("unProject + integral image vs numpy float32: points and all ten planes bit-exact at VGA")
    c.close()


EIG_TOL = 1e-4      # closed-form roots in fp32 (trig of the scaled characteristic polynomial): relative to the largest eigenvalue


def test_gpu_stats_against_numpy_fp32_sums_and_lapack():
    """StatsCalculatorIntegralImage::compute on the GPU (statscalculatorintegralimage.cpp:33-80) without the oracle: numpy float32 integral planes
    (bit-equal to the GPU's, see the test above), getRegion's A + B - C - D in that order (pointintegralimage.cpp:61-64), mean and covariance as
    pointaccumulator.h:66-86 -- every operation fp32 -- give the window count and the mean of EVERY valid pixel bit for bit; the eigenvalues /
    normal of the GPU's closed-form solver (Eigen's computeDirect restated) are then compared with LAPACK on that very fp32 covariance, so the
    noise of the fp32 sums cancels out of the comparison: eigenvalues to 2e-5 of the largest, normals to the solver's accuracy over the eigen-gap."""
    from g2o_frontend_amd import api, synth
    name = "small"
    rows, cols, K, conv, _ = case_params(name)
    c = api.Context(0, rows, cols, 2, omega_storage="exact9")
    proj, converter, _ = gpu_objects(c, name)
    depth = c.DepthImage_convert_16UC1_to_32FC1(synth.render_depth_mm(35, np.eye(4), rows, cols, K))
    g = api.Cloud(c, rows * cols)
    converter.compute(g, depth, keep_stats=True)
    idx, itv = converter.indexImage(), converter.intervalImage()
    a = g.arrays(stats=True)
    f32 = np.float32
    valid = idx >= 0
    P = np.zeros((rows, cols, 3), np.float32); P[valid] = a["points"][:, :3]
    x, y, z = P[..., 0], P[..., 1], P[..., 2]
    planes = [x, y, z, valid.astype(np.float32), x * x, x * y, x * z, y * y, y * z, z * z]
    I = [np.cumsum(np.cumsum(p, axis=1, dtype=np.float32), axis=0, dtype=np.float32) for p in planes]
    rr, cc = np.nonzero(valid)
    rad = np.clip(itv[rr, cc], conv["min_image_radius"], conv["max_image_radius"])
    xmin, xmax = np.clip(cc - rad - 1, 0, cols - 1), np.clip(cc + rad - 1, 0, cols - 1)
    ymin, ymax = np.clip(rr - rad - 1, 0, rows - 1), np.clip(rr + rad - 1, 0, rows - 1)
    v = [((Ik[ymax, xmax] + Ik[ymin, xmin]) - Ik[ymax, xmin]) - Ik[ymin, xmax] for Ik in I]       # pa = I(xmax,ymax); += I(xmin,ymin); -= I(xmin,ymax); -= I(xmax,ymin)
    n = v[3].astype(np.int32)
    has = n >= conv["min_points"]
    assert np.array_equal(a["npoints"], np.where(has, n, 0))
    with np.errstate(divide="ignore", invalid="ignore"):
        d = f32(1.0) / v[3]
    mean = [v[k] * d for k in range(3)]
    gmean = a["stats"][:, 12:15]
    for k in range(3):
        assert np.array_equal(gmean[has, k].view(np.uint32), mean[k][has].astype(np.float32).view(np.uint32)), k
    cov = np.zeros((len(n), 3, 3), np.float64)
    ent = {(0, 0): v[4] * d - mean[0] * mean[0], (1, 0): v[5] * d - mean[1] * mean[0], (2, 0): v[6] * d - mean[2] * mean[0],
           (1, 1): v[7] * d - mean[1] * mean[1], (2, 1): v[8] * d - mean[2] * mean[1], (2, 2): v[9] * d - mean[2] * mean[2]}
    for (i, j), e in ent.items():
        cov[:, i, j] = e; cov[:, j, i] = e
    sel = np.nonzero(has)[0]
    w, V = np.linalg.eigh(cov[sel])
    gev = a["eigenvalues"][sel].astype(np.float64)
    lam = np.abs(w).max(1)
    e0 = np.abs(np.maximum(w[:, 0], 0) - gev[:, 0]) / (lam + 1e-12); e12 = np.abs(w[:, 1:] - gev[:, 1:]).max(1) / (lam + 1e-12)
    print(f"eigenvalues vs LAPACK: worst |d lambda0| / lambda_max {e0.max():.1e}, worst |d lambda1,2| / lambda_max {e12.max():.1e}")
    assert e0.max() <= EIG_TOL and e12.max() <= EIG_TOL
    nrm = a["normals"][sel, :3].astype(np.float64)
    nz = np.abs(nrm).sum(1) > 0
    gap = w[:, 1] - w[:, 0]
    good = nz & (gap > 1e-3 * lam)
    ang = np.arccos(np.clip(np.abs((nrm[good] * V[good, :, 0]).sum(1)), -1, 1))
    tol = 1.5e-4 * lam[good] / gap[good] + 1e-3          # the closed-form solver's accuracy (Eigen documents computeDirect as less accurate than the iterative solver)
    assert good.sum() > 2000 and (ang <= tol).all(), (int(good.sum()), float(ang.max()), float((ang / tol).max()))
    # orientation: towards the sensor (normal . point <= 0)
    assert ((nrm[nz] * a["points"][sel][nz, :3]).sum(1) <= 0).all()
    # what follows the eigen-solve, from the GPU's own eigenvalues / eigenvectors, in numpy with the reference's operations: curvature
    # (stats.h:98-103: fp32 sum, double + 1e-9 and division, rounded to float), the normal's survival (statscalculatorintegralimage.cpp:72-78), the
    # point information matrix U diag U^T of informationmatrixcalculator.cpp:9-36 with its 1 / eigenvalue branch, the class of the normal one -- bit for bit
    ev = a["eigenvalues"][sel]
    U = a["stats"][sel].reshape(-1, 4, 4).transpose(0, 2, 1)[:, :3, :3]                      # column-major 4x4 -> U[:, i, k]
    curv = (ev[:, 0].astype(np.float64) / ((ev[:, 0] + ev[:, 1] + ev[:, 2]).astype(np.float32).astype(np.float64) + 1e-9)).astype(np.float32)
    assert np.array_equal(a["curvature"][sel].view(np.uint32), curv.view(np.uint32))
    keep = curv < f32(conv["stats_curvature_threshold"])
    assert np.array_equal(nz, keep)
    n0 = U[:, :, 0]
    assert np.array_equal(np.abs(a["normals"][sel][keep, :3]).view(np.uint32), np.abs(n0[keep]).view(np.uint32))      # column 0, sign by the flip
    flat = curv < f32(conv["point_info_curvature_threshold"])
    with np.errstate(divide="ignore"):
        dg = np.where(flat[:, None], np.array([1000.0, 1.0, 1.0], np.float32)[None, :], f32(1.0) / ev).astype(np.float32)
    om = np.zeros((len(sel), 3, 3), np.float32)
    for i in range(3):
        for j in range(3):
            om[:, i, j] = ((U[:, i, 0] * dg[:, 0]) * U[:, j, 0] + (U[:, i, 1] * dg[:, 1]) * U[:, j, 1]) + (U[:, i, 2] * dg[:, 2]) * U[:, j, 2]
    gom = a["omega_p"][sel].reshape(-1, 4, 4).transpose(0, 2, 1)[:, :3, :3]
    assert np.array_equal(gom[keep].view(np.uint32), om[keep].view(np.uint32)) and not gom[~keep].any()
    gon = a["omega_n"][sel].reshape(-1, 4, 4).transpose(0, 2, 1)[:, :3, :3]
    want_n = np.where((curv < f32(conv["normal_info_curvature_threshold"]))[:, None, None], np.eye(3, dtype=np.float32) * f32(100.0), np.eye(3, dtype=np.float32))
    assert np.array_equal(gon[keep], want_n[keep]) and not gon[~keep].any()
    print(f"stats vs numpy fp32 + LAPACK: {int(has.sum())} windows (count and mean bit-exact), {int(good.sum())} normals, median angle {np.median(ang):.1e}, worst {ang.max():.1e} rad "
          f"(worst angle / tolerance {float((ang / tol).max()):.2f})")
    c.close()


@pytest.mark.parametrize("name,seed", [("small", 37), ("vga", 5)])
def test_gpu_alignment_against_the_numpy_model(name, seed):
    """Aligner::align (aligner.cpp:49-125) on the GPU against tests/numpy_reference_model.py -- a numpy statement of projector, finder, linearizer and
    Gauss-Newton step written from the reference's source lines, no oracle anywhere: ten iterations, the model leads (the GPU runs every iteration from
    the model's iterate T_i): index images bit for bit, K_i / C_i / inliers_i equal, chi2_i within 1e-5 of the model's float64 sums, and the pose the
    GPU's own 6x6 step arrives at (H + 1001 I, LDL^T, v2t, t2v) within 5e-6 of the model's float64 step."""
    from g2o_frontend_amd import api, synth
    import numpy_reference_model as M
    rows, cols, K, conv, alig = case_params(name)
    c = api.Context(0, rows, cols, 2, omega_storage="exact9")
    proj, converter, aligner = gpu_objects(c, name)
    ref_mm, cur_mm, Ttrue = synth.make_pair(seed, rows, cols, K)
    gr, gc = api.Cloud(c, rows * cols), api.Cloud(c, rows * cols)
    converter.compute(gr, c.DepthImage_convert_16UC1_to_32FC1(ref_mm)); converter.compute(gc, c.DepthImage_convert_16UC1_to_32FC1(cur_mm))
    A, B = gr.arrays(), gc.arrays()
    aligner.setReferenceCloud(gr); aligner.setCurrentCloud(gc); aligner.setOuterIterations(1)
    proj.setImageSize(rows, cols)
    proj.setTransform(np.eye(4, dtype=np.float32))
    cur_index, _ = M.project(B["points"][:, :3], proj.matrices()[0], alig["min_distance"], alig["max_distance"], rows, cols)      # aligner.cpp:60-64
    T = np.eye(4, dtype=np.float32)
    worst_chi2 = worst_step = 0.0
    for it in range(10):
        T[3] = (0, 0, 0, 1)                                                                                   # :72
        proj.setTransform(T)
        ref_index, ref_depth = M.project(A["points"][:, :3], proj.matrices()[0], alig["min_distance"], alig["max_distance"], rows, cols)   # :73-76
        Tinv = api.iso_inverse(T)                                                                             # :79 (Isometry3f::inverse in fp32)
        corr, Kc = M.correspondences(A, B, ref_index, cur_index, Tinv, alig["inlier_normal_angular_threshold"], alig["inlier_distance_threshold"],
                                     alig["flat_curvature_threshold"], alig["inlier_curvature_ratio_threshold"])
        H, b, chi2, inl = M.linearize(A, B, corr, Tinv, alig["inlier_max_chi2"], bool(alig["robust_kernel"]))   # :84-91
        dx = np.linalg.solve(H + 1001.0 * np.eye(6), -b)                                                      # :92-94,110
        invT = api.iso_mul(api.v2t(dx.astype(np.float32)), Tinv)                                              # :111-112
        Tn = api.v2t(api.t2v(api.iso_inverse(invT)))                                                          # :115-116
        # the GPU, one outer iteration from the model's iterate
        aligner.setInitialGuess(T)
        g = aligner.align(images=True)
        f = aligner.correspondenceFinder()
        assert np.array_equal(f.referenceIndexImage(), ref_index) and np.array_equal(f.currentIndexImage(), cur_index), it
        assert np.array_equal(f.referenceDepthImage().view(np.uint32), ref_depth.view(np.uint32)), it
        assert (int(g["K"][0]), int(g["C"][0]), int(g["iter_inliers"][0])) == (Kc, len(corr), inl), (it, g["K"][0], Kc, g["C"][0], len(corr))
        rel = abs(float(g["chi2"][0]) - chi2) / chi2
        assert rel <= CHI2_RTOL, (it, rel)
        step = float(np.abs(g["T"] - Tn).max())
        assert step <= 5e-6, (it, step)
        worst_chi2, worst_step = max(worst_chi2, rel), max(worst_step, step)
        T = Tn.astype(np.float32)
    assert np.abs(T[:3, 3] - Ttrue[:3, 3]).max() < 5e-3                    # the model itself converges to the synthetic motion
    print(f"GPU vs numpy model of Aligner::align ({name}): 10 iterations, index images bit-exact, counters equal, worst chi2 rel diff {worst_chi2:.1e}, "
          f"worst |T_next - model| {worst_step:.1e}")
    c.close()


def test_gpu_depth_helpers_and_match_score_against_the_numpy_model():
    """pwn_static.cpp's depth helpers and PwnMatcherBase::matchClouds' depth-agreement score (pwn_matcher_base.cpp:153-182) on the GPU against
    tests/numpy_reference_model.py (no oracle): conversions and DepthImage_scale bit for bit, the score's counts exactly (the float bitwise '&' with
    the mask included), its mean distance to 1e-6."""
    import ctypes as C
    from g2o_frontend_amd import api, synth
    from g2o_frontend_amd._lib import MatchResult
    import numpy_reference_model as M
    name = "small"
    rows, cols, K, conv, alig = case_params(name)
    big = api.Context(0, 480, 640, 2, omega_storage="exact9")
    raw = synth.render_depth_mm(43, np.eye(4), 480, 640, synth.K_VGA)
    d = big.DepthImage_convert_16UC1_to_32FC1(raw)
    assert np.array_equal(d.view(np.uint32), M.depth_16u_to_32f(raw).view(np.uint32))
    back = d.copy(); back[::7, ::5] = np.finfo(np.float32).max
    assert np.array_equal(big.DepthImage_convert_32FC1_to_16UC1(back), M.depth_32f_to_16u(back))
    for step in (2, 3, 4):
        assert np.array_equal(big.DepthImage_scale(d, step).view(np.uint32), M.depth_scale(d, step).view(np.uint32)), step
    big.close()
    c = api.Context(0, rows, cols, 2, omega_storage="exact9")
    _, converter, aligner = gpu_objects(c, name)
    ref_mm, cur_mm, _ = synth.make_pair(45, rows, cols, K)
    gr, gc = api.Cloud(c, rows * cols), api.Cloud(c, rows * cols)
    converter.compute(gr, c.DepthImage_convert_16UC1_to_32FC1(ref_mm)); converter.compute(gc, c.DepthImage_convert_16UC1_to_32FC1(cur_mm))
    aligner.setReferenceCloud(gr); aligner.setCurrentCloud(gc)
    for outer, thr in ((1, 50.0), (10, 50.0), (3, 5.0)):
        aligner.setOuterIterations(outer)
        aligner.align(images=True)
        f = aligner.correspondenceFinder()
        m = MatchResult()
        c.check(c._L.pwn_hip_match_score(c.h, thr, C.byref(m)))
        w = M.match_score(f.referenceDepthImage(), f.currentDepthImage(), thr)
        assert (m.image_non_zeros, m.image_inliers, m.image_outliers) == (w["image_nonZeros"], w["image_inliers"], w["image_outliers"]), (outer, thr)
        assert abs(m.image_reprojection_distance - w["image_reprojectionDistance"]) <= 1e-6 * w["image_reprojectionDistance"] + 1e-7
    c.close()


@pytest.mark.parametrize("name", ["small", "vga", "kinect"])
def test_gpu_whole_converter_against_the_numpy_model(name):
    """DepthImageConverterIntegralImage::compute on the GPU against the numpy model's complete converter (tests/numpy_reference_model.py: written from
    the reference's source lines, Eigen's computeDirect included; no oracle): every cloud array, the index and the interval image bit for bit -- single
    frame (three-kernel path) and inside a 20-frame batch (single-pass strip kernel + k_stats), synthetic and real Kinect frames."""
    from g2o_frontend_amd import api
    import numpy_reference_model as M
    from test_oracle_vs_numpy_model import _frames, _nz
    rows, cols, K, conv, alig, ref_mm, cur_mm = _frames(name)
    case = "vga" if name == "kinect" else name
    c = api.Context(0, rows, cols, 32, omega_storage="exact9")
    _, converter, _ = gpu_objects(c, case)
    want = [M.convert(M.depth_16u_to_32f(f), K, conv) for f in (ref_mm, cur_mm)]
    g = api.Cloud(c, rows * cols)
    converter.compute(g, c.DepthImage_convert_16UC1_to_32FC1(ref_mm), keep_stats=True)
    a = g.arrays(stats=True)
    assert np.array_equal(converter.indexImage(), want[0]["index"]) and np.array_equal(converter.intervalImage(), want[0]["interval"])
    for k in ("points", "normals", "curvature", "omega_p", "omega_n", "eigenvalues", "npoints"):
        assert np.array_equal(_nz(a[k]), _nz(want[0][k])), (name, k)
    many = [api.Cloud(c, rows * cols) for _ in range(20)]
    converter.computeBatch(many, [ref_mm, cur_mm] * 10, raw_scale=0.001)
    for i in (0, 1, 18, 19):
        b = many[i].arrays()
        for k in ("points", "normals", "curvature", "omega_p", "omega_n"):
            assert np.array_equal(_nz(b[k]), _nz(want[i % 2][k])), (name, i, k)
    c.close()


@pytest.mark.parametrize("kind", [0, 1])
def test_gpu_alignment_with_priors_against_the_numpy_model(kind):
    """Aligner::align with an SE(3) prior (aligner.cpp:96-108, se3_prior.cpp) on the GPU path (pwn_hip_align_with_priors) against the float64 numpy
    model, the model leading: counters equal, the step of H + 1001 I + J^T I' J within 5e-6 (measured 1e-7; the reference's fp32 central differences, eps = 1e-3,
    carry ~1e-4 of relative noise)."""
    from g2o_frontend_amd import api, synth
    import numpy_reference_model as M
    name = "small"
    rows, cols, K, conv, alig = case_params(name)
    c = api.Context(0, rows, cols, 2, omega_storage="exact9")
    proj, converter, aligner = gpu_objects(c, name)
    ref_mm, cur_mm, Ttrue = synth.make_pair(41, rows, cols, K)
    gr, gc = api.Cloud(c, rows * cols), api.Cloud(c, rows * cols)
    converter.compute(gr, c.DepthImage_convert_16UC1_to_32FC1(ref_mm)); converter.compute(gc, c.DepthImage_convert_16UC1_to_32FC1(cur_mm))
    A, B = gr.arrays(), gc.arrays()
    aligner.setReferenceCloud(gr); aligner.setCurrentCloud(gc); aligner.setOuterIterations(1)
    mean = synth.v2t(np.array([0.06, -0.03, -0.02, 0.01, -0.015, 0.01])).astype(np.float32)
    reft = synth.v2t(np.array([0.02, 0.01, -0.01, 0.0, 0.01, 0.0])).astype(np.float32)
    info = (np.diag([4e5, 4e5, 4e5, 2e6, 2e6, 2e6]) + 1e4).astype(np.float32)
    if kind == 0:
        aligner.addRelativePrior(mean, info)
    else:
        aligner.addAbsolutePrior(reft, mean, info)
    cur_index, _ = M.project(B["points"][:, :3], M.projector_matrices(K, np.eye(4, dtype=np.float32))[0], alig["min_distance"], alig["max_distance"], rows, cols)
    T = np.eye(4, dtype=np.float32); worst = 0.0
    for it in range(6):
        T[3] = (0, 0, 0, 1)
        ref_index, _ = M.project(A["points"][:, :3], M.projector_matrices(K, T)[0], alig["min_distance"], alig["max_distance"], rows, cols)
        Tinv = api.iso_inverse(T)
        corr, Kc = M.correspondences(A, B, ref_index, cur_index, Tinv, alig["inlier_normal_angular_threshold"], alig["inlier_distance_threshold"],
                                     alig["flat_curvature_threshold"], alig["inlier_curvature_ratio_threshold"])
        H, b, chi2, inl = M.linearize(A, B, corr, Tinv, alig["inlier_max_chi2"], bool(alig["robust_kernel"]))
        Hp, bp = M.prior_terms(kind, mean, info, Tinv, reft)
        dx = np.linalg.solve(H + 1001.0 * np.eye(6) + Hp, -(b + bp))
        Tn = api.v2t(api.t2v(api.iso_inverse(api.iso_mul(api.v2t(dx.astype(np.float32)), Tinv))))
        aligner.setInitialGuess(T)
        g = aligner.align()
        assert (int(g["K"][0]), int(g["C"][0]), int(g["iter_inliers"][0])) == (Kc, len(corr), inl), it
        assert abs(float(g["chi2"][0]) - chi2) <= CHI2_RTOL * chi2, it
        d = float(np.abs(g["T"] - Tn).max()); worst = max(worst, d)
        assert d <= 5e-6, (it, d)
        T = Tn.astype(np.float32)
    print(f"GPU align with a {'relative' if kind == 0 else 'absolute'} prior vs the numpy model: worst |T_next - model| {worst:.1e}")
    c.close()
