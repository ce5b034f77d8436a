"""Parity on inputs the smooth synthetic room does not produce: sensor-like depth noise, dropouts, rectangular holes, depth steps, exactly
planar and exactly constant patches, out-of-range values -- the branches of the eigensolver (equal eigenvalues, zero scale, the
re-orthogonalisation path), the curvature-ratio test near its bounds and the z-buffer collisions of a noisy cloud.  Bit-exact for the
converter and the index images, the teacher-forced chi2 bar (1e-5) for the linearizer, as in test_gpu_parity.py."""
import numpy as np
import pytest

from conftest import case_params

pytestmark = pytest.mark.gpu

CASES = [("small", s) for s in range(10)] + [("vga", 0), ("vga", 2)]


def disturbed_pair(seed, name="small"):
    from g2o_frontend_amd import synth
    rows, cols, K, _, _ = case_params(name)
    ref_mm, cur_mm, T = synth.make_pair(100 + seed, rows, cols, K)
    rng = np.random.default_rng(1000 + seed)
    out = []
    for mm in (ref_mm, cur_mm):
        z = mm.astype(np.float64) * 1e-3
        sigma = 0.0012 + 0.0019 * (z - 0.4) ** 2                                   # Kinect-like axial noise
        z = z + rng.normal(size=z.shape) * sigma * (1.0 + 3.0 * (seed % 3 == 2))   # every third seed: 4x the noise
        q = np.clip(np.round(z * 1000.0), 0, 65535).astype(np.uint16)
        q[mm == 0] = 0
        q[rng.random(q.shape) < 0.04] = 0                                           # dropouts
        for _ in range(3):                                                          # holes
            r0, c0 = rng.integers(0, rows - 12), rng.integers(0, cols - 12)
            q[r0:r0 + rng.integers(2, 12), c0:c0 + rng.integers(2, 12)] = 0
        r0, c0 = rng.integers(0, rows - 30), rng.integers(0, cols - 40)
        q[r0:r0 + 25, c0:c0 + 35] = 1500                                            # fronto-parallel patch at constant depth: z-variance exactly 0
        r1, c1 = rng.integers(0, rows - 20), rng.integers(0, cols - 20)
        q[r1:r1 + 16, c1:c1 + 16] = (900 + 7 * np.arange(16)[None, :] + 3 * np.arange(16)[:, None]).astype(np.uint16)      # exact plane in mm
        q[rng.integers(0, rows), :] = 7000                                          # beyond max_distance (6 m)
        q[:, rng.integers(0, cols)] = 5                                             # below min_distance after conversion? (5 mm < 0.01 m)
        out.append(q)
    return out[0], out[1], T


@pytest.fixture(scope="module")
def rigs():
    from g2o_frontend_amd import api
    from test_gpu_parity import gpu_objects
    made = {}

    def get(name):
        if name not in made:
            rows, cols, _, _, _ = case_params(name)
            ctx = api.Context(0, rows, cols, 4, omega_storage="exact9")
            _, converter, aligner = gpu_objects(ctx, name)
            made[name] = (ctx, converter, aligner)
        return made[name]
    yield get
    for ctx, _, _ in made.values():
        ctx.close()


@pytest.mark.parametrize("name,seed", CASES)
def test_converter_bit_exact_on_disturbed_frames(rigs, oracle, name, seed):
    from g2o_frontend_amd import api
    from test_gpu_parity import _compare_clouds, oracle_params
    ctx, converter, _ = rigs(name)
    rows, cols, _, _, _ = case_params(name)
    cp, _ = oracle_params(oracle, name)
    ref_mm, cur_mm, _ = disturbed_pair(seed, name)
    for mm in (ref_mm, cur_mm):
        depth = oracle.convert_16u_to_32f(mm)
        oc, oidx, oitv = oracle.convert(cp, depth)
        cloud = api.Cloud(ctx, rows * cols)
        converter.compute(cloud, depth, keep_stats=True)
        assert np.array_equal(oidx, converter.indexImage()) and np.array_equal(oitv, converter.intervalImage())
        o, g = oc.arrays(stats=True), cloud.arrays(stats=True)
        _compare_clouds(o, g, name)
        assert np.array_equal(o["npoints"], g["npoints"])
        assert np.array_equal(o["eigenvalues"].view(np.uint32), g["eigenvalues"].view(np.uint32))
        ok = o["npoints"] > 0
        assert np.array_equal(o["stats"][ok].view(np.uint32), g["stats"][ok].view(np.uint32))
        # the raw uint16 batch path (conversion fused into the kernels, single-pass front end needs >= 16 frames: repeat the frame)
        many = [api.Cloud(ctx, rows * cols) for _ in range(16)]
        ctx.set_subbatch(4, 4)
        converter.computeBatch(many, [mm] * 16, raw_scale=0.001)
        ctx.set_subbatch(64, 64)
        for c in (many[0], many[7], many[15]):
            _compare_clouds(o, c.arrays(), name)


@pytest.mark.parametrize("name,seed", CASES)
def test_alignment_teacher_forced_on_disturbed_frames(rigs, oracle, name, seed):
    """every iteration of the oracle's alignment re-run on the GPU from the oracle's iterate: index images and counters equal, chi2 within 1e-5"""
    from g2o_frontend_amd import api
    from test_gpu_parity import oracle_params
    ctx, converter, aligner = rigs(name)
    rows, cols, K, _, _ = case_params(name)
    cp, ap = oracle_params(oracle, name, accumulate_fp64=1)
    ref_mm, cur_mm, _ = disturbed_pair(seed, name)
    rd, cd = oracle.convert_16u_to_32f(ref_mm), oracle.convert_16u_to_32f(cur_mm)
    oref, _, _ = oracle.convert(cp, rd); ocur, _, _ = oracle.convert(cp, cd)
    o = oracle.align(ap, oref, ocur)
    gref, gcur = api.Cloud(ctx, rows * cols), api.Cloud(ctx, rows * cols)
    converter.compute(gref, rd); converter.compute(gcur, cd)
    aligner.setReferenceCloud(gref); aligner.setCurrentCloud(gcur)
    aligner.setInitialGuess(np.eye(4, dtype=np.float32))
    g = aligner.align(images=True)
    # iteration 0 starts from the same transform on both sides: counters equal, chi2 within the bar
    it0 = o["iterations"][0]
    assert (int(g["K"][0]), int(g["C"][0]), int(g["iter_inliers"][0])) == (it0["K"], it0["C"], it0["inliers"])
    assert abs(float(g["chi2"][0]) - it0["chi2_fp64"]) <= 1e-5 * abs(it0["chi2_fp64"])
    aligner.setOuterIterations(1)
    try:
        for k, it in enumerate(o["iterations"]):
            aligner.setInitialGuess(it["T_before"])
            r = aligner.align(images=True)
            ri = oracle.project(K, it["T_before"], ap.min_distance, ap.max_distance, rows, cols, oref.arrays()["points"])
            f = aligner.correspondenceFinder()._images
            assert np.array_equal(f["ref_index"], ri[0]), f"iteration {k}: reference index image differs"
            assert np.array_equal(f["ref_depth"].view(np.uint32), ri[1].view(np.uint32)), f"iteration {k}: reference depth image differs"
            assert (int(r["K"][0]), int(r["C"][0]), int(r["iter_inliers"][0])) == (it["K"], it["C"], it["inliers"]), k
            assert abs(float(r["chi2"][0]) - it["chi2_fp64"]) <= 1e-5 * abs(it["chi2_fp64"]), (k, r["chi2"][0], it["chi2_fp64"])
    finally:
        aligner.setOuterIterations(10); aligner.setInitialGuess(np.eye(4, dtype=np.float32))
    # free-running: the same pose within the accuracy of two summation orders
    assert np.abs(g["T"] - o["T"]).max() < 2e-3
