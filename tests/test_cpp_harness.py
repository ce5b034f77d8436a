"""The C++ host mirror (g2o_frontend_amd/host/pwn_hip.hpp) driven by the reference-style sequential-odometry harness
(tools/pwn_hip_simple_aligner.cpp ~ pwn_core/pwn_simple_aligner.cpp), against the same harness run through the oracle."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CONF = """// pwn_core/conf/pwn_aligner_1_4.conf values
depthScale 0.001
imageScale 4
fx 525.0
fy 525.0
cx 319.5
cy 239.5
minDistance 0.5
maxDistance 4.5
minImageRadius 3
maxImageRadius 6
minPoints 10
curvatureThreshold 0.2
worldRadius 0.1
informationMatrixCurvatureThreshold 0.02
inlierDistanceThreshold 0.5
inlierNormalAngularThreshold 0.95
inlierCurvatureRatioThreshold 1.3
flatCurvatureThreshold 0.02
inlierMaxChi2 9000
robustKernel 1
outerIterations 10
innerIterations 1
fx 1.0
"""


def write_pgm16(path, img):
    with open(path, "wb") as f:
        f.write(b"P5\n%d %d\n65535\n" % (img.shape[1], img.shape[0]))
        f.write(img.astype(">u2").tobytes())


def test_cpp_mirror_compiles_with_plain_gxx():
    from g2o_frontend_amd import build
    out = build.build_tools(force=True)
    assert os.path.exists(out)


@pytest.mark.gpu
def test_sequential_odometry_harness_matches_oracle(tmp_path, oracle):
    from g2o_frontend_amd import build, synth
    exe = build.build_tools()
    n = 5
    poses = synth.trajectory(7, n)
    frames = [synth.render_depth_mm(7, poses[k], 480, 640, synth.K_VGA, hole_stream=k) for k in range(n)]
    lst = []
    for k, f in enumerate(frames):
        p = tmp_path / f"d{k}.pgm"
        write_pgm16(str(p), f)
        lst.append(f"{k * 0.033:.3f} {p}")
    (tmp_path / "conf.txt").write_text(CONF)
    (tmp_path / "list.txt").write_text("# timestamp file\n" + "\n".join(lst) + "\n")
    subprocess.check_call([exe, str(tmp_path / "conf.txt"), str(tmp_path / "list.txt"), str(tmp_path / "odom.txt")], timeout=300)
    got = np.loadtxt(str(tmp_path / "odom.txt"))
    assert got.shape == (n, 10)
    # the same harness through the oracle (pwn_simple_aligner.cpp:130-183)
    K4 = synth.scaled_K(synth.K_VGA, 4)
    cp = oracle.converter_params(K=K4, **oracle.QVGA4_CONF_CONVERTER)
    ap = oracle.aligner_params(120, 160, K=K4, accumulate_fp64=1, **oracle.QVGA4_CONF_ALIGNER)
    G = np.eye(4, dtype=np.float32)
    prev = None
    for k, f in enumerate(frames):
        d = oracle.depth_scale(oracle.convert_16u_to_32f(f), 4)
        c, _, _ = oracle.convert(cp, d)
        if prev is not None:
            r = oracle.align(ap, prev, c)
            G = (G @ r["T"]).astype(np.float32); G[3] = (0, 0, 0, 1)
            assert abs(got[k, 9] - r["inliers"]) <= 4 and abs(got[k, 8] - r["error"]) <= 1e-2 * r["error"]   # free-running bars
        v = oracle.t2v(G)
        assert np.abs(got[k, 1:4] - v[:3]).max() < 2e-5 and np.abs(got[k, 4:7] - v[3:]).max() < 2e-5, (k, got[k], v)
        prev = c
    # and the chained odometry follows the true camera motion
    true = np.linalg.inv(poses[0]) @ poses[n - 1]
    assert np.abs(G[:3, 3] - true[:3, 3]).max() < 0.02


@pytest.mark.gpu
def test_scene_mapping_harness_matches_oracle(tmp_path, oracle):
    """tools/pwn_hip_scene_aligner.cpp (the mapping loop of pwn_aligner.cpp:150-262 over the C++ mirror: render scene -> sub-scene ->
    align -> Cloud::add -> Merger::merge -> Cloud::save every chunkStep frames) against the same loop run through the oracle.  Both sides
    run free (each on its own poses: a pose that differs in the last bits moves merges and the rendered sub-scene, which feeds back), so the
    comparison is to tolerance: trajectory 1e-3 (a twentieth of the per-frame motion; measured 2.6e-4 after 5 frames at 120x160), scene sizes
    within 0.5 %; the bit-exact version of this loop is tests/test_scene.py::test_incremental_scene_harness; the saved .pwn
    files are read back with the oracle's Cloud::load."""
    from g2o_frontend_amd import build, synth
    build.build_tools()
    exe = os.path.join(ROOT, "tools", "pwn_hip_scene_aligner")
    n = 5
    poses = synth.trajectory(11, n)
    frames = [synth.render_depth_mm(11, poses[k], 480, 640, synth.K_VGA, hole_stream=k) for k in range(n)]
    lst = []
    for k, f in enumerate(frames):
        p = tmp_path / f"d{k}.pgm"
        write_pgm16(str(p), f)
        lst.append(f"{k * 0.033:.3f} {p}")
    (tmp_path / "conf.txt").write_text(CONF + "chunkStep 3\n")
    (tmp_path / "list.txt").write_text("\n".join(lst) + "\n")
    prefix = str(tmp_path / "run")
    subprocess.check_call([exe, str(tmp_path / "conf.txt"), str(tmp_path / "list.txt"), prefix], timeout=300)
    got = np.loadtxt(prefix + "_trajectory.txt")
    assert got.shape == (n, 8)
    # the same loop through the oracle
    K4 = synth.scaled_K(synth.K_VGA, 4)
    conv, alig = oracle.QVGA4_CONF_CONVERTER, oracle.QVGA4_CONF_ALIGNER
    cp = oracle.converter_params(K=K4, **conv)
    ap = oracle.aligner_params(120, 160, K=K4, accumulate_fp64=1, **alig)
    G = np.eye(4, dtype=np.float32); S = np.eye(4, dtype=np.float32)
    scene = oracle.Cloud(); counter = 0; saved = {}
    oracle.set_gaussians(True)
    try:
        for k, f in enumerate(frames):
            d = oracle.depth_scale(oracle.convert_16u_to_32f(f), 4)
            c, _, _ = oracle.convert(cp, d)
            if k > 0:
                _, rendered = oracle.project(K4, S, conv["min_distance"], conv["max_distance"], 120, 160, scene.arrays()["points"])
                sub, _, _ = oracle.convert(cp, rendered)
                r = oracle.align(ap, sub, c)
                G = oracle.iso_mul(G, r["T"]); G[3] = (0, 0, 0, 1)
                S = oracle.iso_mul(S, r["T"]); S[3] = (0, 0, 0, 1)
                c0 = counter; counter += 1
                if c0 % 3 == 0:
                    saved[counter] = len(scene)
                    S = np.eye(4, dtype=np.float32); scene = oracle.Cloud()
            scene.add(c, S)
            oracle.merge(scene, K4, S, conv["min_distance"], conv["max_distance"], 120, 160)
            v = oracle.t2v(G)
            assert np.abs(got[k, 1:4] - v[:3]).max() < 1e-3 and np.abs(got[k, 4:7] - v[3:]).max() < 1e-3, (k, got[k], v)
            assert abs(got[k, 7] - len(scene)) <= max(5, 0.005 * len(scene)), (k, got[k, 7], len(scene))
        saved[counter] = len(scene)
    finally:
        oracle.set_gaussians(False)
    # every scene file the harness wrote: readable by the oracle's loader, point count as reported / as the oracle's own loop
    for cnt, size in saved.items():
        path = f"{prefix}_scene-{cnt:03d}.pwn"
        assert os.path.exists(path), path
        cl, T = oracle.Cloud.load(path)
        assert cl is not None and abs(len(cl) - size) <= max(5, 0.005 * size), (cnt, len(cl), size)
        a = cl.arrays()
        assert np.isfinite(a["points"]).all() and np.abs(np.linalg.norm(a["normals"][:, :3], axis=1)[np.abs(a["normals"][:, :3]).sum(1) > 0] - 1).max() < 1e-3
    true = np.linalg.inv(poses[0]) @ poses[n - 1]
    assert np.abs(G[:3, 3] - true[:3, 3]).max() < 0.02


def _run_tracker_app(tmp_path, n=5, extra_conf="", prefix_name="run"):
    """tools/pwn_hip_tracker_app on a 5-frame VGA stream at matcher scale 4: every frame a key-frame (fraction 2.0), a cache of two clouds"""
    from g2o_frontend_amd import build, synth
    build.build_tools()
    exe = os.path.join(ROOT, "tools", "pwn_hip_tracker_app")
    poses = synth.trajectory(13, n)
    frames = [synth.render_depth_mm(13, poses[k], 480, 640, synth.K_VGA, hole_stream=k) for k in range(n)]
    lst = []
    for k, f in enumerate(frames):
        p = tmp_path / f"d{k}.pgm"
        write_pgm16(str(p), f)
        lst.append(f"{k * 0.033:.3f} {p}")
    # every frame becomes a keyframe (inliers fraction is always < 2), a cache of two clouds forces misses and evictions
    (tmp_path / "conf.txt").write_text(CONF + "newFrameInliersFraction 2.0\ncacheSize 2\nframeMaxOutliersThreshold 100000\n" + extra_conf)
    (tmp_path / "list.txt").write_text("\n".join(lst) + "\n")
    prefix = str(tmp_path / prefix_name)
    subprocess.check_call([exe, str(tmp_path / "conf.txt"), str(tmp_path / "list.txt"), prefix], timeout=300)
    track = np.loadtxt(prefix + "_track.txt")
    clos = np.loadtxt(prefix + "_closures.txt", comments="#", ndmin=2)
    hits_line = [l for l in open(prefix + "_closures.txt") if l.startswith("#")][0].split()
    extras = {l.split()[0]: np.array(l.split()[1:], np.float64) for l in open(prefix + "_extras.txt")}
    assert track.shape == (n, 22) and clos.shape == (n * (n - 1) // 2, 24)
    return n, poses, frames, track, clos, hits_line, extras


@pytest.mark.gpu
def test_tracker_closer_harness_matches_oracle(tmp_path, oracle):
    """The same C++ program against the ORACLE running the same flow (not against the other mirror): PwnTracker::processFrame
    (pwn_tracker/pwn_tracker.cpp:106-215) through tests/oracle_tracker.py, the closure pass as matchClouds (pwn_matcher_base.cpp:88-183: guess
    with z reset, align, depth-agreement score) + matchFrames' acceptance (pwn_closer.cpp:138-141), _computeStatistics, priors.
    Free-running bars of the 120x160 configuration: counters within a few flipped correspondences, poses 5e-5."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from g2o_frontend_amd import synth
    from oracle_tracker import OracleTracker
    from test_statistics import omega_tolerance
    O = oracle
    n, poses, frames, track, clos, hits_line, extras = _run_tracker_app(tmp_path)
    conv, alig = O.QVGA4_CONF_CONVERTER, O.QVGA4_CONF_ALIGNER
    K = synth.K_VGA
    Kmat = np.array([[K[0], 0, K[2]], [0, K[1], K[3]], [0, 0, 1]], np.float32)
    I = np.eye(4, dtype=np.float32)
    otr = OracleTracker(O, conv, alig, 4, 2.0)
    depth = [O.convert_16u_to_32f(f) for f in frames]
    keyPoses, clouds = [], []
    for k in range(n):
        o = otr.processFrame(depth[k], I, Kmat)
        g = track[k]
        assert (int(g[0]), bool(g[1])) == (k, o["newFrame"])
        if k > 0:
            assert bool(g[2]) and abs(int(g[3]) - o["inliers"]) <= 4 and abs(g[4] - o["error"]) <= 1e-2 * o["error"], (k, g[:6], o)
        assert np.abs(g[6:].astype(np.float32).reshape(4, 4).T - o["globalT"]).max() < 5e-5, k
        keyPoses.append(o["globalT"]); clouds.append(otr.prev)
    K4 = synth.scaled_K(K, 4)
    acc_nonzero, acc_outliers, acc_inliers = 3000, 100000, 1000                   # pwn_closer.cpp:56-58 with the harness' outlier threshold
    row, accepted = 0, 0
    for a in range(n):
        for b in range(a):
            guess = O.iso_mul(O.iso_inverse(keyPoses[b]), keyPoses[a]).astype(np.float32)
            guess[2, 3] = 0.0                                                        # pwn_matcher_base.cpp:114
            ap = O.aligner_params(120, 160, K=K4, initial_guess=guess, accumulate_fp64=1, **alig)
            o = O.align(ap, clouds[b], clouds[a], images=True)
            sc = O.match_score(o["ref_depth"], o["cur_depth"], 50.0)
            g = clos[row]; row += 1
            assert (int(g[0]), int(g[1])) == (b, a)
            assert abs(int(g[3]) - o["inliers"]) <= 8, (g[:8], o["inliers"])
            assert abs(int(g[4]) - sc["image_nonZeros"]) <= 16 and abs(int(g[6]) - sc["image_inliers"]) <= 16 and abs(int(g[5]) - sc["image_outliers"]) <= 16, (g[:8], sc)
            assert abs(g[7] - sc["image_reprojectionDistance"]) <= 2e-2 * sc["image_reprojectionDistance"] + 1e-3
            assert np.abs(g[8:].astype(np.float32).reshape(4, 4).T - o["T"]).max() < 1e-4       # the guesses are the two trackers' own key poses (5e-5 apart)
            ok = not (sc["image_nonZeros"] < acc_nonzero or sc["image_outliers"] > acc_outliers or sc["image_inliers"] < acc_inliers)
            margin = min(abs(sc["image_nonZeros"] - acc_nonzero), abs(sc["image_inliers"] - acc_inliers))
            if margin > 16:
                assert bool(g[2]) == ok, (g[:8], sc)
            accepted += int(bool(g[2]))
    assert accepted >= 1
    # Aligner::_computeStatistics on frames 0 -> 1 (identity guess), per-entry bars measured on the oracle's own chain (test_statistics.omega_tolerance)
    ap = O.aligner_params(120, 160, K=K4, accumulate_fp64=1, **alig)
    o = O.align(ap, clouds[0], clouds[1])
    os_ = O.align_statistics(ap, clouds[0], clouds[1], o["T"])
    e = extras["statistics"]
    Hg = e[39:75].astype(np.float32).reshape(6, 6).T; omg = e[3:39].astype(np.float32).reshape(6, 6).T
    assert np.abs(e[75:91].astype(np.float32).reshape(4, 4).T - o["T"]).max() < 5e-5
    dH = np.abs(Hg - os_["H"]).max() / np.abs(os_["H"]).max()
    assert dH <= 2e-2
    tol_om, tol_ratio = omega_tolerance(O, os_["H"], o["T"], dH)
    assert (np.abs(omg.astype(np.float64) - os_["omega"]) <= tol_om).all()
    assert abs(e[1] - os_["translationalEigenRatio"]) <= tol_ratio[0] and abs(e[2] - os_["rotationalEigenRatio"]) <= tol_ratio[1]
    # two priors (absolute + relative, both identity / 1000 I) on the same pair
    O.clear_priors(); O.add_prior(1, I, np.eye(6, dtype=np.float32) * 1000, reference_transform=I); O.add_prior(0, I, np.eye(6, dtype=np.float32) * 1000)
    try:
        op = O.align(ap, clouds[0], clouds[1])
    finally:
        O.clear_priors()
    e = extras["priors"]
    assert int(e[0]) == 2 and abs(int(e[1]) - op["inliers"]) <= 4 and abs(e[2] - op["error"]) <= 1e-2 * op["error"]
    assert np.abs(e[3:19].astype(np.float32).reshape(4, 4).T - op["T"]).max() < 5e-5
    assert np.abs(op["T"] - o["T"]).max() > 1e-5                              # the priors did change the estimate


@pytest.mark.gpu
def test_tracker_closer_harness_matches_python_mirror(tmp_path):
    """tools/pwn_hip_tracker_app.cpp -- PwnTracker::processFrame with key-cloud switching, the closure pass through CloudCache + batched
    matchClouds + PwnCloser's acceptance rule, Aligner statistics (omega, eigen ratios) and SE(3) priors, all over the C++ mirror --
    against the same flow written with the Python mirror.  Both are thin layers over the same C-ABI calls with the same arguments, so
    every number must agree to the last digit printed (%.9g round-trips a float)."""
    from g2o_frontend_amd import api, synth
    from oracle import oracle as O          # parameter tables only
    n, poses, frames, track, clos, hits_line, extras = _run_tracker_app(tmp_path)

    # the same flow with the Python mirror
    conv, alig = O.QVGA4_CONF_CONVERTER, O.QVGA4_CONF_ALIGNER
    K = synth.K_VGA
    Kmat = np.array([[K[0], 0, K[2]], [0, K[1], K[3]], [0, 0, 1]], np.float32)
    ctx = api.Context(0, 480, 640, 32)
    try:
        cproj, aproj = api.PinholePointProjector(), api.PinholePointProjector()
        for p in (cproj, aproj):
            p.setMinDistance(conv["min_distance"]); p.setMaxDistance(conv["max_distance"])
        stats = api.StatsCalculatorIntegralImage()
        stats.setWorldRadius(conv["world_radius"]); stats.setMinImageRadius(conv["min_image_radius"]); stats.setMaxImageRadius(conv["max_image_radius"])
        stats.setMinPoints(conv["min_points"]); stats.setCurvatureThreshold(conv["stats_curvature_threshold"])
        pinfo, ninfo = api.PointInformationMatrixCalculator(), api.NormalInformationMatrixCalculator()
        pinfo.setCurvatureThreshold(conv["point_info_curvature_threshold"]); ninfo.setCurvatureThreshold(conv["normal_info_curvature_threshold"])
        converter = api.DepthImageConverterIntegralImage(cproj, stats, pinfo, ninfo)
        finder = api.CorrespondenceFinder()
        finder.setInlierDistanceThreshold(alig["inlier_distance_threshold"]); finder.setInlierNormalAngularThreshold(alig["inlier_normal_angular_threshold"])
        finder.setFlatCurvatureThreshold(alig["flat_curvature_threshold"]); finder.setInlierCurvatureRatioThreshold(alig["inlier_curvature_ratio_threshold"])
        lin = api.Linearizer(); lin.setInlierMaxChi2(alig["inlier_max_chi2"]); lin.setRobustKernel(alig["robust_kernel"])
        aligner = api.Aligner(ctx)
        aligner.setProjector(aproj); aligner.setLinearizer(lin); aligner.setCorrespondenceFinder(finder)
        aligner.setOuterIterations(alig["outer_iterations"]); aligner.setInnerIterations(alig["inner_iterations"])
        tracker = api.PwnTracker(aligner, converter); tracker.setScale(4); tracker.setNewFrameInliersFraction(2.0)
        I = np.eye(4, dtype=np.float32)
        depth = [np.where(f > 0, np.float32(0.001) * f.astype(np.float32), np.float32(0)).astype(np.float32) for f in frames]
        keyPoses = []
        for k in range(n):
            out = tracker.processFrame(depth[k], I, Kmat)
            g = track[k]
            assert (int(g[0]), bool(g[1]), bool(g[2]), int(g[3])) == (k, out["newFrame"], out["aligned"], out["inliers"]), (k, g[:6], out)
            assert np.float32(g[4]) == np.float32(out["error"]) and np.float32(g[5]) == np.float32(out.get("inliersFraction", 0.0))
            assert np.array_equal(g[6:].astype(np.float32).reshape(4, 4).T, out["globalT"]), (k, g[6:], out["globalT"])
            assert out["newFrame"]
            keyPoses.append(out["globalT"])
        # chained odometry follows the true camera motion
        true = np.linalg.inv(poses[0]) @ poses[n - 1]
        assert np.abs(keyPoses[-1][:3, 3] - true[:3, 3]).max() < 0.02
        cache = api.CloudCache(tracker, capacity=2)
        for k in range(n):
            cache.addFrame(k, depth[k], Kmat, I)
        acc = api.PwnCloserAcceptance(frameMaxOutliersThreshold=100000)
        row = 0
        for a in range(n):
            for b in range(a):
                cur = cache.get(a); oth = cache.get(b)
                guess = api.iso_mul(api.iso_inverse(keyPoses[b]), keyPoses[a])
                m = tracker.matchCloudsBatch([oth], [cur], I, I, Kmat, 480, 640, [guess])[0]
                g = clos[row]; row += 1
                assert (int(g[0]), int(g[1])) == (b, a)
                assert (bool(g[2]), int(g[3]), int(g[4]), int(g[5]), int(g[6])) == (acc.accept(m), m["cloud_inliers"], m["image_nonZeros"], m["image_outliers"], m["image_inliers"]), (g[:8], m)
                assert np.float32(g[7]) == np.float32(m["image_reprojectionDistance"])
                assert np.array_equal(g[8:].astype(np.float32).reshape(4, 4).T, m["transform"].astype(np.float32))
                # the closure transform agrees with the tracked relative pose (same scene, consistent estimates)
                assert np.abs(m["transform"][:3, 3] - guess[:3, 3]).max() < 0.02
        assert (int(hits_line[3]), int(hits_line[5])) == (cache.hits, cache.misses) and cache.misses > n
        # the same pass through the multi-GPU form inside the app (flat cloud of `current`: export -> host buffer -> import; results as 288-byte
        # records: Cloud::exportFlat / importFlat, PwnMatcherBase::matchCloudsBatchRecords of the C++ mirror): equal to the plain calls, record for record
        rec_line = [l for l in open(str(tmp_path / "run") + "_closures.txt") if l.startswith("# replica_record_calls")][0].split()
        assert int(rec_line[2]) == n * (n - 1) // 2 and int(rec_line[4]) == 1, rec_line
        assert clos[:, 2].sum() >= 1                 # some closure is accepted
        # statistics and priors
        ca, cb = cache.get(0), cache.get(1)
        aproj.setCameraMatrix(Kmat); aproj.setImageSize(480, 640); aproj.scale(np.float32(0.25))
        finder.setImageSize(aproj.imageRows(), aproj.imageCols())
        aligner.setSensorOffset(I); aligner.setInitialGuess(I)
        aligner.setReferenceCloud(ca); aligner.setCurrentCloud(cb)
        r = aligner.align(statistics=True)
        e = extras["statistics"]
        assert bool(e[0]) == aligner.solutionValid()
        assert np.float32(e[1]) == np.float32(aligner.translationalEigenRatio()) and np.float32(e[2]) == np.float32(aligner.rotationalEigenRatio())
        assert np.array_equal(e[3:39].astype(np.float32), aligner.omega().ravel(order="F"))
        assert np.array_equal(e[39:75].astype(np.float32), lin._H.ravel(order="F"))
        assert np.array_equal(e[75:91].astype(np.float32).reshape(4, 4).T, r["T"])
        info = np.eye(6, dtype=np.float32) * 1000
        aligner.setReferenceCloud(ca); aligner.setCurrentCloud(cb)
        aligner.addAbsolutePrior(I, I, info); aligner.addRelativePrior(I, info)
        rp = aligner.align()
        e = extras["priors"]
        assert (int(e[0]), int(e[1])) == (2, rp["inliers"]) and np.float32(e[2]) == np.float32(rp["error"])
        assert np.array_equal(e[3:19].astype(np.float32).reshape(4, 4).T, rp["T"])
        assert not np.array_equal(rp["T"], r["T"])       # the priors did change the estimate
        aligner.clearPriors()
        # stage-level calls
        aproj.setTransform(I)
        ri, _ = aproj.project(ca); ci, _ = aproj.project(cb)
        aligner.setReferenceCloud(ca); aligner.setCurrentCloud(cb)
        corr, ncand = aligner.computeCorrespondences(ri, ci, I)
        l = aligner.linearize(corr, I)
        e = extras["stages"]
        assert (int(e[0]), int(e[1]), int(e[2])) == (len(corr), ncand, l["inliers"]) and np.float32(e[3]) == np.float32(l["chi2"]) and int(e[4]) == int(ri.astype(np.int64).sum())
        assert np.array_equal(e[5:41].astype(np.float32), l["H"].ravel(order="F")) and np.array_equal(e[41:47].astype(np.float32), l["b"])
        scaled = O.depth_scale(depth[0], 4)
        pts = api.Cloud(ctx, scaled.size)
        ui = aproj.unProject(pts, scaled)
        itv = aproj.projectIntervals(ctx, scaled, conv["world_radius"])
        e = extras["projector"]
        assert (int(e[0]), int(e[1]), int(e[2])) == (pts.size(), int(ui.astype(np.int64).sum()), int(itv.astype(np.int64).sum()))
    finally:
        ctx.close()


@pytest.mark.gpu
def test_cpp_bench_matches_python_mirror(tmp_path):
    """tools/pwn_hip_bench.cpp (bench.py's step -- batched convert of the raw frames resident in HBM + batched align -- driven from C++)
    returns, pair for pair, the bits the Python mirror returns for the same frames and parameters."""
    from g2o_frontend_amd import api, build, synth
    from test_gpu_parity import gpu_objects
    build.build_tools()
    exe = os.path.join(ROOT, "tools", "pwn_hip_bench")
    npairs = 3
    pairs = [synth.make_pair(500 + i, 480, 640, synth.K_VGA) for i in range(npairs)]
    names = []
    for i, (r, c, _) in enumerate(pairs):
        for tag, img in (("r", r), ("c", c)):
            p = tmp_path / f"{tag}{i}.pgm"
            write_pgm16(str(p), img); names.append(str(p))
    (tmp_path / "list.txt").write_text("\n".join(names) + "\n")
    out = subprocess.check_output([exe, str(tmp_path / "list.txt"), "32", "2", "1"], timeout=300).decode().splitlines()
    head = out[0].split()
    assert head[0] == "pairs" and int(head[1]) == 32 and float(head[7]) > 100.0, out[0]
    rows = [np.array(l.split()[1:], np.float64) for l in out[1:] if l.startswith("pair")]
    assert len(rows) == npairs
    # mode 3: the same step as ONE submission (Aligner::convertAlignBatch of the C++ mirror): the same lines, character for character
    one = subprocess.check_output([exe, str(tmp_path / "list.txt"), "32", "2", "1", "0", "3"], timeout=300).decode().splitlines()
    assert [l for l in one[1:] if l.startswith("pair")] == [l for l in out[1:] if l.startswith("pair")] and one[0].split()[-1] == "3"
    ctx = api.Context(0, 480, 640, 8)
    try:
        _, converter, aligner = gpu_objects(ctx, "vga")
        refs = [api.Cloud(ctx, 480 * 640) for _ in pairs]; curs = [api.Cloud(ctx, 480 * 640) for _ in pairs]
        converter.computeBatch(refs + curs, [p[0] for p in pairs] + [p[1] for p in pairs], raw_scale=0.001)
        res = aligner.alignBatch(refs, curs)
        for g, r in zip(rows, res):
            assert int(g[1]) == r["inliers"] and np.float32(g[2]) == np.float32(r["error"])
            assert np.array_equal(g[3:19].astype(np.float32).reshape(4, 4).T, r["T"])
            assert np.array_equal(g[19:].astype(np.float32), r["chi2"])
    finally:
        ctx.close()


@pytest.mark.gpu
def test_tracker_app_with_look_ahead_writes_the_same_files(tmp_path):
    """`lookAhead 1` in the configuration: PwnTracker::prefetch hands frame k+1 to the library's helper thread before frame k is aligned
    (pwn_hip_convert_scaled_begin / _end).  Track, closure and extras files are byte-identical to the plain run's."""
    _run_tracker_app(tmp_path, prefix_name="plain")
    _run_tracker_app(tmp_path, extra_conf="lookAhead 1\n", prefix_name="ahead")
    for suffix in ("_track.txt", "_closures.txt", "_extras.txt"):
        assert (tmp_path / ("plain" + suffix)).read_bytes() == (tmp_path / ("ahead" + suffix)).read_bytes(), suffix
