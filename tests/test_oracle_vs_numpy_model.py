"""The ORACLE (oracle/pwn_oracle.cpp, the checker of the GPU parity tests) against tests/numpy_reference_model.py -- numpy written from the reference's
source lines, sharing no code with the oracle.  Runs on the CPU.  Integer results and everything built from single fp32 operations in the reference's
order must agree bit for bit; the closed-form eigen-solve and the least-squares sums to a stated tolerance.  (tests/test_gpu_parity.py holds the GPU
against the same model.)"""
import numpy as np
import pytest

from conftest import case_params
import numpy_reference_model as M

f32 = np.float32


@pytest.fixture(scope="module")
def scene(oracle):
    from g2o_frontend_amd import synth
    rows, cols, K, conv, alig = case_params("small")
    ref_mm, cur_mm, Ttrue = synth.make_pair(41, rows, cols, K)
    ref, cur = oracle.convert_16u_to_32f(ref_mm), oracle.convert_16u_to_32f(cur_mm)
    cp = oracle.converter_params(K=K, **conv)
    oref, ridx, ritv = oracle.convert(cp, ref); ocur, cidx, _ = oracle.convert(cp, cur)
    return dict(rows=rows, cols=cols, K=K, conv=conv, alig=alig, ref=ref, cur=cur, oref=oref, ocur=ocur, ridx=ridx, ritv=ritv, cidx=cidx, Ttrue=Ttrue, cp=cp)


def test_projector_matrices_bit_equal(oracle):
    from g2o_frontend_amd import synth
    rng = np.random.default_rng(1)
    for K in (synth.K_VGA, synth.scaled_K(synth.K_VGA, 4), synth.K_1280):
        for _ in range(20):
            T = synth.v2t(np.concatenate([rng.uniform(-1, 1, 3), rng.uniform(-0.3, 0.3, 3)])).astype(np.float32)
            for a, b in zip(oracle.projector_matrices(K, T), M.projector_matrices(K, T)):
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_converter_front_end_and_window_statistics(oracle, scene):
    s = scene; conv = s["conv"]; rows, cols = s["rows"], s["cols"]
    _, iKRt, _ = M.projector_matrices(s["K"], np.eye(4, dtype=np.float32))
    valid, x, y, z = M.unproject(s["ref"], iKRt, conv["min_distance"], conv["max_distance"])
    a = s["oref"].arrays(stats=True)
    assert np.array_equal(s["ridx"] >= 0, valid) and np.array_equal(s["ridx"][valid], np.arange(valid.sum()))
    for k, img in enumerate((x, y, z)):
        assert np.array_equal(a["points"][:, k].view(np.uint32), img[valid].view(np.uint32))
    assert np.array_equal(s["ritv"], M.intervals(s["ref"], valid, s["K"], conv["world_radius"]))
    I = M.integral_planes(valid, x, y, z)
    oI = oracle.integral_image(s["ridx"], a["points"])
    for k in range(10):
        assert np.array_equal(oI[k].view(np.uint32), I[k].view(np.uint32)), k
    v = M.window_sums(I, s["ritv"], valid, conv["min_image_radius"], conv["max_image_radius"])
    n, mean, cov = M.mean_and_covariance(v)
    has = n >= conv["min_points"]
    assert np.array_equal(a["npoints"], np.where(has, n, 0))
    for k in range(3):
        assert np.array_equal(a["stats"][has, 12 + k].view(np.uint32), mean[k][has].astype(np.float32).view(np.uint32)), k
    # eigen-solve against LAPACK on the same fp32 covariance; what follows it bit for bit from the oracle's own eigen data
    C3 = np.zeros((len(n), 3, 3)); 
    for (i, j), e in cov.items():
        C3[:, i, j] = e; C3[:, j, i] = e
    sel = np.nonzero(has)[0]
    w, V = np.linalg.eigh(C3[sel]); lam = np.abs(w).max(1) + 1e-12
    ev = a["eigenvalues"][sel]
    assert (np.abs(np.maximum(w[:, 0], 0) - ev[:, 0]) / lam).max() < 1e-4 and (np.abs(w[:, 1:] - ev[:, 1:]).max(1) / lam).max() < 1e-4
    curv = (ev[:, 0].astype(np.float64) / ((ev[:, 0] + ev[:, 1] + ev[:, 2]).astype(np.float32).astype(np.float64) + 1e-9)).astype(np.float32)
    assert np.array_equal(a["curvature"][sel].view(np.uint32), curv.view(np.uint32))
    keep = curv < f32(conv["stats_curvature_threshold"])
    nrm = a["normals"][sel, :3]
    assert np.array_equal(np.abs(nrm).sum(1) > 0, keep)
    U = a["stats"][sel].reshape(-1, 4, 4).transpose(0, 2, 1)[:, :3, :3]
    assert np.array_equal(np.abs(nrm[keep]).view(np.uint32), np.abs(U[keep, :, 0]).view(np.uint32))
    assert ((nrm[keep].astype(np.float64) * a["points"][sel][keep, :3]).sum(1) <= 0).all()
    gap = w[:, 1] - w[:, 0]; good = keep & (gap > 1e-3 * lam)
    ang = np.arccos(np.clip(np.abs((nrm[good].astype(np.float64) * V[good, :, 0]).sum(1)), -1, 1))
    assert (ang <= 1.5e-4 * lam[good] / gap[good] + 1e-3).all() and np.median(ang) < 1e-4
    flat = curv < f32(conv["point_info_curvature_threshold"])
    with np.errstate(divide="ignore"):
        dg = np.where(flat[:, None], np.array([1000.0, 1.0, 1.0], np.float32)[None, :], f32(1.0) / ev).astype(np.float32)
    om = np.zeros((len(sel), 3, 3), np.float32)
    for i in range(3):
        for j in range(3):
            om[:, i, j] = ((U[:, i, 0] * dg[:, 0]) * U[:, j, 0] + (U[:, i, 1] * dg[:, 1]) * U[:, j, 1]) + (U[:, i, 2] * dg[:, 2]) * U[:, j, 2]
    gom = a["omega_p"][sel].reshape(-1, 4, 4).transpose(0, 2, 1)[:, :3, :3]
    assert np.array_equal(gom[keep].view(np.uint32), om[keep].view(np.uint32)) and not gom[~keep].any()


def test_projector_and_finder_exact(oracle, scene):
    from g2o_frontend_amd import synth
    s = scene; alig = s["alig"]; rows, cols = s["rows"], s["cols"]
    A, B = s["oref"].arrays(), s["ocur"].arrays()
    ap = oracle.aligner_params(rows, cols, K=s["K"], **alig)
    cur_index, _ = M.project(B["points"][:, :3], M.projector_matrices(s["K"], np.eye(4, dtype=np.float32))[0], alig["min_distance"], alig["max_distance"], rows, cols)
    for v in ([0.0] * 6, [0.03, -0.02, 0.05, 0.01, -0.015, 0.02], [-0.2, 0.1, 0.3, -0.05, 0.04, 0.03]):
        T = synth.v2t(np.array(v)).astype(np.float32); T[3] = (0, 0, 0, 1)
        wi, wd = M.project(A["points"][:, :3], M.projector_matrices(s["K"], T)[0], alig["min_distance"], alig["max_distance"], rows, cols)
        oi, od = oracle.project(s["K"], T, alig["min_distance"], alig["max_distance"], rows, cols, A["points"])
        assert np.array_equal(oi, wi) and np.array_equal(od.view(np.uint32), wd.view(np.uint32)), v
        Tinv = oracle.iso_inverse(T)
        ocorr, oK = oracle.correspondences(ap, s["oref"], s["ocur"], wi, cur_index, Tinv)
        mcorr, mK = M.correspondences(A, B, wi, cur_index, Tinv, alig["inlier_normal_angular_threshold"], alig["inlier_distance_threshold"],
                                      alig["flat_curvature_threshold"], alig["inlier_curvature_ratio_threshold"])
        assert oK == mK and np.array_equal(ocorr, mcorr), v


def test_alignment_teacher_forced_by_the_model(oracle, scene):
    """Aligner::align (aligner.cpp:49-125), the model leads: every iteration of the oracle from the model's iterate -- counters equal, chi2 within 1e-5
    of the model's float64 sums, the oracle's own step (fp32 H, LDL^T, v2t, t2v) within 5e-6 of the model's float64 step."""
    s = scene; alig = s["alig"]; rows, cols = s["rows"], s["cols"]
    A, B = s["oref"].arrays(), s["ocur"].arrays()
    cur_index, _ = M.project(B["points"][:, :3], M.projector_matrices(s["K"], np.eye(4, dtype=np.float32))[0], alig["min_distance"], alig["max_distance"], rows, cols)
    T = np.eye(4, dtype=np.float32)
    for it in range(10):
        T[3] = (0, 0, 0, 1)
        ref_index, _ = M.project(A["points"][:, :3], M.projector_matrices(s["K"], T)[0], alig["min_distance"], alig["max_distance"], rows, cols)
        Tinv = oracle.iso_inverse(T)
        corr, Kc = M.correspondences(A, B, ref_index, cur_index, Tinv, alig["inlier_normal_angular_threshold"], alig["inlier_distance_threshold"],
                                     alig["flat_curvature_threshold"], alig["inlier_curvature_ratio_threshold"])
        H, b, chi2, inl = M.linearize(A, B, corr, Tinv, alig["inlier_max_chi2"], bool(alig["robust_kernel"]))
        dx = np.linalg.solve(H + 1001.0 * np.eye(6), -b)
        Tn = oracle.v2t(oracle.t2v(oracle.iso_inverse(oracle.iso_mul(oracle.v2t(dx.astype(np.float32)), Tinv))))
        ap = oracle.aligner_params(rows, cols, K=s["K"], initial_guess=T, accumulate_fp64=1, **dict(alig, outer_iterations=1))
        o = oracle.align(ap, s["oref"], s["ocur"])
        i0 = o["iterations"][0]
        assert (i0["K"], i0["C"], i0["inliers"]) == (Kc, len(corr), inl), it
        assert abs(i0["chi2_fp64"] - chi2) <= 1e-5 * chi2, (it, i0["chi2_fp64"], chi2)
        assert np.abs(o["T"] - Tn).max() <= 5e-6, it
        T = Tn.astype(np.float32)
    assert np.abs(T[:3, 3] - s["Ttrue"][:3, 3]).max() < 5e-3


def test_depth_helpers_and_match_score(oracle, scene):
    from g2o_frontend_amd import synth
    s = scene; rows, cols = s["rows"], s["cols"]
    raw = synth.render_depth_mm(43, np.eye(4), 480, 640, synth.K_VGA)
    d = oracle.convert_16u_to_32f(raw)
    assert np.array_equal(d.view(np.uint32), M.depth_16u_to_32f(raw).view(np.uint32))
    back = d.copy(); back[::7, ::5] = np.finfo(np.float32).max
    assert np.array_equal(oracle.convert_32f_to_16u(back), M.depth_32f_to_16u(back))
    for step in (2, 3, 4):
        assert np.array_equal(oracle.depth_scale(d, step).view(np.uint32), M.depth_scale(d, step).view(np.uint32)), step
    # the score on the finder's depth images of two poses (FLT_MAX where nothing projects)
    A = s["oref"].arrays(); B = s["ocur"].arrays(); alig = s["alig"]
    _, cd = M.project(B["points"][:, :3], M.projector_matrices(s["K"], np.eye(4, dtype=np.float32))[0], alig["min_distance"], alig["max_distance"], rows, cols)
    for v in ([0.0] * 6, [0.02, -0.01, 0.04, 0.01, 0.0, -0.01]):
        T = synth.v2t(np.array(v)).astype(np.float32)
        _, rd = M.project(A["points"][:, :3], M.projector_matrices(s["K"], T)[0], alig["min_distance"], alig["max_distance"], rows, cols)
        o, m = oracle.match_score(rd, cd, 50.0), M.match_score(rd, cd, 50.0)
        assert (o["image_nonZeros"], o["image_inliers"], o["image_outliers"]) == (m["image_nonZeros"], m["image_inliers"], m["image_outliers"]), v
        assert abs(o["image_reprojectionDistance"] - m["image_reprojectionDistance"]) <= 1e-6 * abs(m["image_reprojectionDistance"]) + 1e-7, v


def _nz(a):
    """bits with the sign of zero dropped (the reference's 4x4 products add exact-zero fourth terms: -0 + 0 = +0)"""
    a = np.ascontiguousarray(a)
    if a.dtype != np.float32:
        return a
    a = a.copy(); a[a == 0] = 0
    return a.view(np.uint32)


def _frames(name):
    import os
    from g2o_frontend_amd import synth
    if name == "kinect":
        z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kinect_real_pair.npz"))
        rows, cols, K, conv, alig = case_params("vga")
        return rows, cols, K, conv, alig, z["ref_mm"], z["cur_mm"]
    rows, cols, K, conv, alig = case_params(name)
    ref_mm, cur_mm, _ = synth.make_pair(3, rows, cols, K)
    return rows, cols, K, conv, alig, ref_mm, cur_mm


@pytest.mark.parametrize("name", ["small", "vga", "kinect"])
def test_whole_converter_bit_for_bit(oracle, name):
    """DepthImageConverterIntegralImage::compute: every array the oracle produces against the numpy model's -- points, index and interval images,
    window counts, eigenvalues (Eigen's computeDirect restated in numpy fp32), normals, curvature, both information matrices -- bit for bit, on a
    synthetic frame at 120x160 and VGA and on a real Kinect frame of the reference repository (25 000 points on the 1 / eigenvalue branch of
    informationmatrixcalculator.cpp:27-29)."""
    rows, cols, K, conv, alig, ref_mm, _ = _frames(name)
    depth = oracle.convert_16u_to_32f(ref_mm)
    o, oidx, oitv = oracle.convert(oracle.converter_params(K=K, **conv), depth)
    a = o.arrays(stats=True)
    m = M.convert(depth, K, conv)
    assert np.array_equal(oidx, m["index"]) and np.array_equal(oitv, m["interval"])
    for k in ("points", "normals", "curvature", "omega_p", "omega_n", "eigenvalues", "npoints"):
        assert np.array_equal(_nz(a[k]), _nz(m[k])), (name, k)
    if name == "kinect":
        flat = a["curvature"] < 0.02
        assert int(((np.abs(a["normals"][:, :3]).sum(1) > 0) & ~flat).sum()) > 10000      # the non-flat branch is really exercised


def test_compute_statistics_against_the_float64_model(oracle, scene):
    """Aligner::_computeStatistics (aligner.cpp:152-199, unscented.h): the oracle's fp32 chain (JacobiSVD solve, LLT, sigma points, 6x6 inverse) and the
    product's host function (pwn_hip_compute_statistics, no GPU needed) against the float64 numpy model on the H of real alignments and on disturbed
    copies: mean 1e-6, omega 2e-4 of its largest entry, the two eigen ratios 1e-3."""
    import ctypes as C
    from g2o_frontend_amd import _lib
    s = scene; rows, cols = s["rows"], s["cols"]
    ap = oracle.aligner_params(rows, cols, K=s["K"], accumulate_fp64=1, **s["alig"])
    o = oracle.align(ap, s["oref"], s["ocur"])
    st = oracle.align_statistics(ap, s["oref"], s["ocur"], o["T"])
    rng = np.random.default_rng(2)
    for k in range(6):
        H = st["H"].astype(np.float64)
        if k:
            E = rng.uniform(-1, 1, (6, 6)); H = H * (1.0 + 0.05 * (E + E.T) / 2) + np.diag(rng.uniform(0, 0.1, 6) * np.diag(H))
        H = H.astype(np.float32)
        m = M.compute_statistics(H, o["T"])
        mean = np.empty(6, np.float32); om = np.empty(36, np.float32); tr, rr = C.c_float(0), C.c_float(0)
        Hc = np.ascontiguousarray(H.T.reshape(-1), np.float32); Tc = np.ascontiguousarray(o["T"].T.reshape(-1), np.float32)
        _lib.lib().pwn_hip_compute_statistics(Hc.ctypes.data_as(C.c_void_p), Tc.ctypes.data_as(C.c_void_p), mean.ctypes.data_as(C.c_void_p),
                                             om.ctypes.data_as(C.c_void_p), C.byref(tr), C.byref(rr))
        prod = dict(mean=mean, omega=om.reshape(6, 6).T, translationalEigenRatio=tr.value, rotationalEigenRatio=rr.value)
        for got in (oracle.compute_statistics(H, o["T"]), prod):
            assert np.abs(got["mean"] - m["mean"]).max() < 1e-6, k
            assert np.abs(got["omega"] - m["omega"]).max() <= 2e-4 * np.abs(m["omega"]).max(), k
            assert abs(got["translationalEigenRatio"] / m["translationalEigenRatio"] - 1) < 1e-3 and abs(got["rotationalEigenRatio"] / m["rotationalEigenRatio"] - 1) < 1e-3, k


@pytest.mark.parametrize("kind", [0, 1])
def test_alignment_with_priors_teacher_forced_by_the_model(oracle, scene, kind):
    """Aligner::align with an SE(3) prior (aligner.cpp:96-108, se3_prior.cpp:8-71), the model leads: every iteration of the oracle from the model's
    iterate; the step of H + 1001 I + J^T I' J against the float64 model's (the reference's fp32 central differences with eps = 1e-3 carry ~1e-4 of
    relative noise in J, hence the 5e-5 bar on the pose)."""
    from g2o_frontend_amd import synth
    s = scene; alig = s["alig"]; rows, cols = s["rows"], s["cols"]
    A, B = s["oref"].arrays(), s["ocur"].arrays()
    mean = synth.v2t(np.array([0.06, -0.03, -0.02, 0.01, -0.015, 0.01])).astype(np.float32)
    reft = synth.v2t(np.array([0.02, 0.01, -0.01, 0.0, 0.01, 0.0])).astype(np.float32)
    info = (np.diag([4e5, 4e5, 4e5, 2e6, 2e6, 2e6]) + 1e4).astype(np.float32)
    oracle.clear_priors()
    oracle.add_prior(kind, mean, info, reference_transform=reft if kind else None)
    try:
        cur_index, _ = M.project(B["points"][:, :3], M.projector_matrices(s["K"], np.eye(4, dtype=np.float32))[0], alig["min_distance"], alig["max_distance"], rows, cols)
        T = np.eye(4, dtype=np.float32)
        worst = 0.0
        for it in range(6):
            T[3] = (0, 0, 0, 1)
            ref_index, _ = M.project(A["points"][:, :3], M.projector_matrices(s["K"], T)[0], alig["min_distance"], alig["max_distance"], rows, cols)
            Tinv = oracle.iso_inverse(T)
            corr, Kc = M.correspondences(A, B, ref_index, cur_index, Tinv, alig["inlier_normal_angular_threshold"], alig["inlier_distance_threshold"],
                                         alig["flat_curvature_threshold"], alig["inlier_curvature_ratio_threshold"])
            H, b, chi2, inl = M.linearize(A, B, corr, Tinv, alig["inlier_max_chi2"], bool(alig["robust_kernel"]))
            Hp, bp = M.prior_terms(kind, mean, info, Tinv, reft)
            dx = np.linalg.solve(H + 1001.0 * np.eye(6) + Hp, -(b + bp))
            Tn = oracle.v2t(oracle.t2v(oracle.iso_inverse(oracle.iso_mul(oracle.v2t(dx.astype(np.float32)), Tinv))))
            ap = oracle.aligner_params(rows, cols, K=s["K"], initial_guess=T, accumulate_fp64=1, **dict(alig, outer_iterations=1))
            o = oracle.align(ap, s["oref"], s["ocur"])
            i0 = o["iterations"][0]
            assert (i0["K"], i0["C"], i0["inliers"]) == (Kc, len(corr), inl), it
            worst = max(worst, float(np.abs(o["T"] - Tn).max()))
            assert np.abs(o["T"] - Tn).max() <= 5e-5, (it, np.abs(o["T"] - Tn).max())
            T = Tn.astype(np.float32)
        # the prior is felt: the iterate sits between the free solution and the prior's mean
        assert np.abs(T[:3, 3] - s["Ttrue"][:3, 3]).max() > 2e-3
    finally:
        oracle.clear_priors()
