#!/usr/bin/env python3
"""Generates tests/golden/tracker_vga_sweep.json: BASELINE configs[2] at its stated size -- the 200-frame synthetic VGA stream
synth.trajectory_sweep(9, 200) through PwnTracker::processFrame semantics (pwn_tracker/pwn_tracker.cpp:106-215), matcher scale 1,
reference configuration pwn_aligner_1_1.conf -- run through the CPU oracle (tests/oracle_tracker.py).  Per frame: key-frame decision,
inliers, chi2, globalT (after the re-orthonormalisation of every 50th frame, pwn_tracker.cpp:154-159), the aligner's T.  About 90 s on 8 cores.  The GPU test (tests/test_tracker.py) compares all 200 frames with this file and re-runs the
oracle live on a prefix to show the file is what the oracle gives."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE))); sys.path.insert(0, os.path.dirname(HERE))
from conftest import case_params  # noqa: E402
from g2o_frontend_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402
from oracle_tracker import OracleTracker  # noqa: E402

SEED, FRAMES, FRACTION, SCALE = 9, 200, 0.4, 1


def main():
    rows, cols, K, conv, alig = case_params("vga")
    otr = OracleTracker(O, conv, alig, SCALE, FRACTION)
    Km = np.array([[K[0], 0, K[2]], [0, K[1], K[3]], [0, 0, 1]], np.float32)
    I = np.eye(4, dtype=np.float32)
    poses = synth.trajectory_sweep(SEED, FRAMES)
    rec = []
    for k in range(FRAMES):
        depth = O.convert_16u_to_32f(synth.render_depth_mm(SEED, poses[k], rows, cols, K, hole_stream=k))
        o = otr.processFrame(depth, I, Km)
        rec.append(dict(newFrame=bool(o["newFrame"]), inliers=int(o.get("inliers", 0)), error=float(o.get("error", 0.0)),
                        globalT=[float(np.float32(v)) for v in o["globalT"].reshape(-1)],
                        T=[float(np.float32(v)) for v in o["T"].reshape(-1)] if "T" in o else None))
        if k % 20 == 0:
            print(k, rec[-1]["newFrame"], rec[-1]["inliers"], file=sys.stderr, flush=True)
    out = dict(seed=SEED, frames=FRAMES, scale=SCALE, newFrameInliersFraction=FRACTION, rows=rows, cols=cols,
               trajectory="synth.trajectory_sweep(9, 200)", keyframes=[k for k, r in enumerate(rec) if r["newFrame"]], per_frame=rec)
    with open(os.path.join(HERE, "tracker_vga_sweep.json"), "w") as f:
        json.dump(out, f)
    print("keyframes", out["keyframes"])


if __name__ == "__main__":
    main()
