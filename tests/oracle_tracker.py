"""PwnTracker::processFrame (pwn_tracker/pwn_tracker.cpp:106-215) restated on top of the CPU oracle: the test-side twin of
g2o_frontend_amd.api.PwnTracker, used by tests/test_tracker.py and by tests/golden/make_tracker_golden.py.  Test infrastructure only."""
import numpy as np


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
            if not (self.counter % 50):                       # pwn_tracker.cpp:154-159: every 50th frame R <- R - 0.5 R (R^T R - I)
                self.globalT = O.reorthonormalize(self.globalT)
            self.globalT[3] = (0, 0, 0, 1)
            out.update(T=res["T"].copy(), inliers=res["inliers"], error=res["error"])
            if np.float32(res["inliers"]) / np.float32(r * c) < self.fraction:
                out["newFrame"] = True; self.keyframes += 1
                self.prev, self.prevT = cur, self.globalT.copy()
        else:
            out["newFrame"] = True; self.prev, self.prevT, self.prevOff = cur, self.globalT.copy(), np.asarray(off, np.float32).copy()
            self.keyframes += 1
        self.counter += 1
        out["globalT"] = self.globalT.copy()
        return out
