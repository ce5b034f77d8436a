"""PwnTracker::processFrame key-cloud logic (pwn_tracker/pwn_tracker.cpp:106-215) and the
device-resident cloud cache (pwn_tracker/pwn_tracker_cache.cpp:24-51), against the same harness run through the oracle."""
import numpy as np
import pytest

from conftest import case_params

pytestmark = pytest.mark.gpu


from oracle_tracker import OracleTracker  # noqa: E402  (processFrame restated on top of the oracle)

# bars of test_config2_full_size_stream_matches_oracle; measured on MI355X (printed by the test, profiles/r03_parity_measured.txt):
#   free-running, 200 frames / 5 key-cloud switches: worst |inliers diff| 6 of ~150 000, worst |globalT diff| 5.5e-6
#   teacher-forced (one alignment from the oracle's own state): inliers equal on 158 of 199 frames, worst diff 4, worst |T diff| 2.0e-6
# An inlier difference is a correspondence at a pixel border of the z-buffer or at a threshold of the finder: the ten iterations of an
# alignment run free on both sides and their iterates differ in the last bits (summation order of H, b); from the SAME iterate the
# counters are equal (tests/test_gpu_parity.py: teacher-forced per iteration).
FREE_INLIERS_BAR, FREE_POSE_BAR = 16, 2e-5
FORCED_INLIERS_BAR, FORCED_POSE_BAR = 8, 1e-5


def test_processFrame_trajectory_and_keyframes_match_oracle(oracle):
    from g2o_frontend_amd import api, synth
    from test_gpu_parity import gpu_objects
    rows, cols, K, _, _ = case_params("vga")
    _, _, _, conv, alig = case_params("small")
    ctx = api.Context(0, rows, cols, 2)
    _, converter, aligner = gpu_objects(ctx, "small")
    tracker = api.PwnTracker(aligner, converter); tracker.setScale(4); tracker.setNewFrameInliersFraction(0.66)
    otr = OracleTracker(oracle, conv, alig, 4, 0.66)
    Km = np.array([[K[0], 0, K[2]], [0, K[1], K[3]], [0, 0, 1]], np.float32)
    off = synth.v2t(np.array([0.0, 0.0, 0.0, 0.01, -0.005, 0.0])).astype(np.float32)
    n = 9
    poses = synth.trajectory(3, n)
    switches = 0
    for k in range(n):
        depth = oracle.convert_16u_to_32f(synth.render_depth_mm(3, poses[k], rows, cols, K, hole_stream=k))
        g = tracker.processFrame(depth, off, Km)
        o = otr.processFrame(depth, off, Km)
        assert g["newFrame"] == o["newFrame"], k
        if k > 0:
            assert abs(g["inliers"] - o["inliers"]) <= 4, (k, g["inliers"], o["inliers"])
        assert np.abs(g["globalT"] - o["globalT"]).max() < 5e-5, (k, np.abs(g["globalT"] - o["globalT"]).max())
        switches += int(g["newFrame"])
    assert tracker.numKeyframes() == otr.keyframes == switches and 1 < switches < n
    # the chained pose follows the true motion (sensor offset frame)
    true = off.astype(np.float64) @ (np.linalg.inv(poses[0]) @ poses[n - 1]) @ np.linalg.inv(off.astype(np.float64))
    assert np.abs(tracker.globalT()[:3, 3] - true[:3, 3]).max() < 0.03
    ctx.close()


def test_cloud_cache_lru_and_reconversion(oracle):
    from g2o_frontend_amd import api, synth
    from test_gpu_parity import gpu_objects
    rows, cols, K, _, _ = case_params("vga")
    ctx = api.Context(0, rows, cols, 2)
    _, converter, aligner = gpu_objects(ctx, "small")
    matcher = api.PwnMatcherBase(aligner, converter); matcher.setScale(4)
    cache = api.CloudCache(matcher, capacity=2)
    Km = np.array([[K[0], 0, K[2]], [0, K[1], K[3]], [0, 0, 1]], np.float32)
    I = np.eye(4, dtype=np.float32)
    ref = {}
    for k in range(3):
        d = oracle.convert_16u_to_32f(synth.render_depth_mm(40 + k, np.eye(4), rows, cols, K))
        cache.addFrame(k, d, Km, I)
        ref[k] = cache.get(k).arrays()["points"].copy()
    assert (cache.hits, cache.misses) == (0, 3)
    assert np.array_equal(cache.get(2).arrays()["points"], ref[2]) and cache.hits == 1          # resident
    assert np.array_equal(cache.get(0).arrays()["points"], ref[0]) and cache.misses == 4        # evicted -> converted again, same bits
    cache.get(2); cache.get(1)
    assert cache.misses == 5 and cache.hits == 2
    ctx.close()


def test_config2_full_size_stream_matches_oracle():
    """BASELINE configs[2] at its stated size: the 200-frame VGA stream of synth.trajectory_sweep(9) through PwnTracker::processFrame
    (pwn_tracker/pwn_tracker.cpp:106-215) at matcher scale 1 with the reference's new-frame fraction 0.4 -- a trajectory that switches the
    key-cloud five times (:164-185) and passes the re-orthonormalisation of every 50th frame (:154-159) three times.  Every frame against the
    oracle tracker's record of the same stream (tests/golden/tracker_vga_sweep.json, made by tests/golden/make_tracker_golden.py):

    * free-running leg: the tracker runs on its own state from frame 0: key-frame decisions identical, inliers and globalT within the
      free-running bars accumulated along the chain of key-frames;
    * teacher-forced leg: a second tracker whose chain state (_globalT, _previousCloudTransform; its key cloud is bit-identical by the
      converter's parity) is set to the oracle's before every frame, so each alignment starts from the oracle's own initial guess and only
      the ten iterations of ONE alignment separate the two sides: inliers within a handful, aligner T and globalT at the single-alignment bar;
    * the oracle is re-run live on a prefix that contains the first switch and, from the recorded state, across frame 50 (bit-identical to
      the file: the file is the oracle's output, re-orthonormalisation included);
    * size-independent properties on all 200 frames."""
    import json
    import os
    from g2o_frontend_amd import api, synth
    from test_gpu_parity import gpu_objects
    from oracle import oracle as O
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tracker_vga_sweep.json")))
    n, rows, cols = gold["frames"], gold["rows"], gold["cols"]
    assert (n, rows, cols, gold["scale"]) == (200, 480, 640, 1)
    _, _, K, conv, alig = case_params("vga")
    ctx = api.Context(0, rows, cols, 2)
    _, converter, aligner = gpu_objects(ctx, "vga")
    alproj = api.PinholePointProjector(); alproj.setMinDistance(alig["min_distance"]); alproj.setMaxDistance(alig["max_distance"])
    aligner.setProjector(alproj)
    tracker = api.PwnTracker(aligner, converter); tracker.setScale(1); tracker.setNewFrameInliersFraction(gold["newFrameInliersFraction"])
    forced = api.PwnTracker(aligner, converter); forced.setScale(1); forced.setNewFrameInliersFraction(gold["newFrameInliersFraction"])
    otr = OracleTracker(O, conv, alig, 1, gold["newFrameInliersFraction"])
    Km = np.array([[K[0], 0, K[2]], [0, K[1], K[3]], [0, 0, 1]], np.float32)
    I = np.eye(4, dtype=np.float32)
    poses = synth.trajectory_sweep(gold["seed"], n)
    gT = [np.asarray(e["globalT"], np.float32).reshape(4, 4) for e in gold["per_frame"]]
    live_prefix = 30                                    # the oracle costs ~0.45 s per VGA frame; the first switch is at frame 24
    live_again = (49, 52)                               # ... and frames 49..51 from the recorded state: the step of pwn_tracker.cpp:154-159 at frame 50
    keyframes, worst_pose, worst_inl = [], 0.0, 0
    f_worst_pose, f_worst_T, f_worst_inl, f_equal = 0.0, 0.0, 0, 0
    last_key = 0
    key_depth = None
    for k in range(n):
        depth = ctx.DepthImage_convert_16UC1_to_32FC1(synth.render_depth_mm(gold["seed"], poses[k], rows, cols, K, hole_stream=k))
        e = gold["per_frame"][k]
        # --- the oracle, live
        if k == live_again[0]:                          # restart the oracle tracker from the file's state just before frame 49
            otr = OracleTracker(O, conv, alig, 1, gold["newFrameInliersFraction"])
            otr.prev, _, _, _ = otr.makeCloud(Km, I, key_depth)
            otr.prevT, otr.prevOff, otr.globalT, otr.counter = gT[last_key].copy(), I.copy(), gT[k - 1].copy(), k
        if k < live_prefix or live_again[0] <= k < live_again[1]:
            o = otr.processFrame(depth, I, Km)
            assert o["newFrame"] == e["newFrame"] and int(o.get("inliers", 0)) == e["inliers"], k
            assert np.array_equal(o["globalT"].reshape(-1).astype(np.float32), np.asarray(e["globalT"], np.float32)), k
        # --- free-running leg
        g = tracker.processFrame(depth, I, Km)
        assert g["newFrame"] == e["newFrame"], (k, g.get("inliers"), e["inliers"])
        if g["newFrame"]:
            keyframes.append(k)
        if k > 0:
            worst_inl = max(worst_inl, abs(g["inliers"] - e["inliers"]))
            assert abs(g["inliers"] - e["inliers"]) <= FREE_INLIERS_BAR, (k, g["inliers"], e["inliers"])         # of ~150 000
            assert g["inliers"] > 0 and g["aligned"]
        dT = np.abs(g["globalT"].reshape(-1) - np.asarray(e["globalT"], np.float32)).max()
        worst_pose = max(worst_pose, float(dT))
        assert dT < FREE_POSE_BAR, (k, dT)              # free-running chain: 1e-5-class per alignment, chained over the key-frames
        # --- teacher-forced leg
        if k > 0:
            forced._globalT = gT[k - 1].copy(); forced._previousCloudTransform = gT[last_key].copy()
        f = forced.processFrame(depth, I, Km)
        assert f["newFrame"] == e["newFrame"], (k, f.get("inliers"), e["inliers"])
        if k > 0:
            d_inl = abs(f["inliers"] - e["inliers"])
            f_worst_inl = max(f_worst_inl, d_inl); f_equal += int(d_inl == 0)
            assert d_inl <= FORCED_INLIERS_BAR, (k, f["inliers"], e["inliers"])
            dTa = float(np.abs(f["T"].reshape(-1) - np.asarray(e["T"], np.float32).reshape(4, 4).reshape(-1)).max())
            dTg = float(np.abs(f["globalT"].reshape(-1) - np.asarray(e["globalT"], np.float32)).max())
            f_worst_T, f_worst_pose = max(f_worst_T, dTa), max(f_worst_pose, dTg)
            assert dTa < FORCED_POSE_BAR and dTg < FORCED_POSE_BAR, (k, dTa, dTg)
        if e["newFrame"]:
            last_key, key_depth = k, depth
        # --- properties that hold at any size: globalT is a rigid transform and follows the true camera motion
        R = g["globalT"][:3, :3].astype(np.float64)
        assert np.abs(R.T @ R - np.eye(3)).max() < 1e-4 and np.array_equal(g["globalT"][3], [0, 0, 0, 1])
        true = np.linalg.inv(poses[0]) @ poses[k]
        assert np.abs(g["globalT"][:3, 3] - true[:3, 3]).max() < 0.04, (k, g["globalT"][:3, 3], true[:3, 3])   # odometry drift; the oracle's own chain reaches 2.1 cm
    assert keyframes == gold["keyframes"] and len(keyframes) - 1 >= 3          # >= 3 key-cloud switches after the first frame
    assert tracker.numKeyframes() == forced.numKeyframes() == len(keyframes)
    print(f"config2: key-frames {keyframes}; free-running: worst |inliers diff| {worst_inl}, worst |globalT diff| {worst_pose:.2e}; "
          f"teacher-forced: inliers equal on {f_equal} of {n - 1} frames, worst |diff| {f_worst_inl}, worst |T diff| {f_worst_T:.2e}, "
          f"worst |globalT diff| {f_worst_pose:.2e}")
    ctx.close()


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_makeCloud_in_two_halves_is_makeCloud(oracle):
    """pwn_hip_convert_scaled_begin / _end: the cloud of the helper thread is the cloud of pwn_hip_convert_scaled, bit for bit, while the
    context runs an alignment in between; the misuse cases return errors instead of racing."""
    from g2o_frontend_amd import api, synth
    from test_gpu_parity import gpu_objects
    rows, cols, K, _, _ = case_params("vga")
    ctx = api.Context(0, rows, cols, 2)
    _, converter, aligner = gpu_objects(ctx, "vga")
    m = api.PwnMatcherBase(aligner, converter); m.setScale(1)
    Km = np.array([[K[0], 0, K[2]], [0, K[1], K[3]], [0, 0, 1]], np.float32)
    I = np.eye(4, dtype=np.float32)
    poses = synth.trajectory_sweep(9, 4)
    frames = [oracle.convert_16u_to_32f(synth.render_depth_mm(9, poses[k], rows, cols, K, hole_stream=k)) for k in range(4)]
    plain = [m.makeCloud(Km, I, f) for f in frames]
    aligner.setReferenceCloud(plain[0][0]); aligner.setCurrentCloud(plain[1][0]); aligner.setInitialGuess(I)
    base = aligner.align()
    for k, f in enumerate(frames):
        t = m.makeCloudBegin(Km, I, f)
        between = aligner.align()                                  # the context is busy while the helper converts
        cloud, r, c, Ks = m.makeCloudEnd(t)
        assert (r, c) == plain[k][1:3] and np.array_equal(Ks, plain[k][3])
        assert cloud.size() == plain[k][0].size() > rows * cols // 3
        mine, ref = cloud.arrays(), plain[k][0].arrays()
        for key in ref:
            assert np.array_equal(_bits(mine[key]), _bits(ref[key])), (k, key)
        assert np.array_equal(_bits(between["T"]), _bits(base["T"])) and np.array_equal(_bits(between["chi2"]), _bits(base["chi2"]))
        # the cloud made by the helper serves as an alignment's current cloud like any other
        aligner.setCurrentCloud(cloud); g = aligner.align(); aligner.setCurrentCloud(plain[k][0]); h = aligner.align()
        assert np.array_equal(_bits(g["T"]), _bits(h["T"])) and np.array_equal(_bits(g["chi2"]), _bits(h["chi2"]))
        aligner.setCurrentCloud(plain[1][0])
    # misuse: a second begin while one is in flight; end of a cloud that was never begun; a frame larger than the context
    t = m.makeCloudBegin(Km, I, frames[0])
    with pytest.raises(api.PwnHipError, match="already in flight"):
        m.makeCloudBegin(Km, I, frames[1])
    with pytest.raises(api.PwnHipError, match="no conversion of this cloud"):
        ctx.check(ctx._L.pwn_hip_convert_end(ctx.h, plain[0][0].h))
    m.makeCloudEnd(t)
    with pytest.raises(api.PwnHipError):
        m.makeCloudBegin(Km, I, np.zeros((rows + 8, cols), np.float32))
    # a conversion error surfaces at the end: a cloud too small for the frame's valid pixels
    small = api.Cloud(ctx, 1000)
    p = converter.params(I)
    ctx.check(ctx._L.pwn_hip_convert_scaled_begin(ctx.h, api.C.byref(p), api._ptr(frames[0]), rows, cols, 1, 0.01, small.h))
    with pytest.raises(api.PwnHipError, match="capacity"):
        ctx.check(ctx._L.pwn_hip_convert_end(ctx.h, small.h))
    # destroying the cloud, or the context, with a conversion in flight waits for it
    t = m.makeCloudBegin(Km, I, frames[2]); t["cloud"].__del__()
    t = m.makeCloudBegin(Km, I, frames[3])
    ctx.close()


def test_tracker_with_prefetched_frames_is_the_tracker(oracle):
    """PwnTracker.prefetch / processFrame(nextDepthImage=...): frame k+1 converted next to the alignment of frame k -- every frame's
    result bitwise the plain tracker's, key-cloud switches included; a prefetched frame that is not the next one is dropped."""
    from g2o_frontend_amd import api, synth
    from test_gpu_parity import gpu_objects
    rows, cols, K, _, _ = case_params("vga")
    ctx = api.Context(0, rows, cols, 2)
    _, converter, aligner = gpu_objects(ctx, "small")
    Km = np.array([[K[0], 0, K[2]], [0, K[1], K[3]], [0, 0, 1]], np.float32)
    off = synth.v2t(np.array([0.0, 0.0, 0.0, 0.01, -0.005, 0.0])).astype(np.float32)
    n = 12
    poses = synth.trajectory(3, n)
    frames = [oracle.convert_16u_to_32f(synth.render_depth_mm(3, poses[k], rows, cols, K, hole_stream=k)) for k in range(n)]

    def run(mode):
        tr = api.PwnTracker(aligner, converter); tr.setScale(4); tr.setNewFrameInliersFraction(0.66)
        out = []
        if mode == "prefetch":
            tr.prefetch(frames[0], off, Km)
        for k, f in enumerate(frames):
            nxt = frames[k + 1] if k + 1 < n else None
            if mode == "plain":
                out.append(tr.processFrame(f, off, Km))
            elif mode == "next":
                out.append(tr.processFrame(f, off, Km, nextDepthImage=nxt))
            else:
                if k == 5:
                    tr.prefetch(frames[0], off, Km)             # not the frame that comes next: processFrame must not take it
                out.append(tr.processFrame(f, off, Km))
                if nxt is not None and k != 4:
                    tr.prefetch(nxt, off, Km)
        tr.dropPrefetched()
        return out, tr.numKeyframes()

    plain, kf = run("plain")
    assert 1 < kf < n
    for mode in ("next", "prefetch"):
        got, kf2 = run(mode)
        assert kf2 == kf
        for k, (a, b) in enumerate(zip(plain, got)):
            assert a["newFrame"] == b["newFrame"] and a["inliers"] == b["inliers"], (mode, k)
            assert np.array_equal(_bits(a["globalT"]), _bits(b["globalT"])), (mode, k)
            if k > 0:
                assert np.array_equal(_bits(a["T"]), _bits(b["T"])) and _bits(np.float32(a["error"])) == _bits(np.float32(b["error"])), (mode, k)
    ctx.close()
