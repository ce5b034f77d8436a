"""PwnTracker::processFrame key-cloud logic (pwn_tracker/pwn_tracker.cpp:106-215) and the
device-resident cloud cache (pwn_tracker/pwn_tracker_cache.cpp:24-51), against the same harness run through the oracle."""
import numpy as np
import pytest

from conftest import case_params

pytestmark = pytest.mark.gpu


class OracleTracker:
    """processFrame restated on top of the oracle (test-side twin of g2o_frontend_amd.api.PwnTracker)."""

    def __init__(self, O, conv, alig, scale, fraction):
        self.O, self.conv, self.alig, self.scale, self.fraction = O, conv, alig, scale, fraction
        self.prev = None
        I = np.eye(4, dtype=np.float32)
        self.globalT, self.prevT, self.prevOff = I.copy(), I.copy(), I.copy()
        self.counter = 0; self.keyframes = 0

    def makeCloud(self, K, off, depth):
        O = self.O
        Ks = (np.asarray(K, np.float32) * (np.float32(1.0) / np.float32(self.scale))).astype(np.float32); Ks[2, 2] = 1
        k4 = (float(Ks[0, 0]), float(Ks[1, 1]), float(Ks[0, 2]), float(Ks[1, 2]))
        d = O.depth_scale(depth, self.scale)
        c, _, _ = O.convert(O.converter_params(K=k4, sensor_offset=off, **self.conv), d)
        return c, d.shape[0], d.shape[1], k4

    def processFrame(self, depth, off, K):
        O = self.O
        cur, r, c, k4 = self.makeCloud(K, off, depth)
        out = dict(newFrame=False)
        if self.prev is not None:
            guess = O.iso_mul(O.iso_mul(O.iso_inverse(self.prevT), self.globalT), np.eye(4, dtype=np.float32))
            ap = O.aligner_params(r, c, K=k4, initial_guess=guess, reference_sensor_offset=self.prevOff, current_sensor_offset=off,
                                  accumulate_fp64=1, **self.alig)
            res = O.align(ap, self.prev, cur)
            self.globalT = O.iso_mul(self.prevT, res["T"]) if res["inliers"] > 0 else O.iso_mul(self.globalT, guess)
            self.globalT[3] = (0, 0, 0, 1)
            out.update(inliers=res["inliers"], error=res["error"])
            if np.float32(res["inliers"]) / np.float32(r * c) < self.fraction:
                out["newFrame"] = True; self.keyframes += 1
                self.prev, self.prevT = cur, self.globalT.copy()
        else:
            out["newFrame"] = True; self.prev, self.prevT, self.prevOff = cur, self.globalT.copy(), np.asarray(off, np.float32).copy()
            self.keyframes += 1
        self.counter += 1
        out["globalT"] = self.globalT.copy()
        return out


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
