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
