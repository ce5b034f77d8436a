"""CPU tests of the oracle (the restatement of the reference path) and of the synthetic generator.

The reference cannot be built here and holds no golden vectors ("parity unpinned"), so the oracle is pinned
the only ways available: (1) every Eigen routine it restates is checked against an independent float64
implementation (numpy / scipy); (2) the algebra it follows (window sums, Jacobian of the error function,
SE(3) conventions) is checked against brute-force / finite-difference evaluations; (3) the whole chain must
recover the known camera motion and the known plane normals of the synthetic scene; (4) committed golden
vectors (tests/golden/) freeze its outputs bit for bit.
"""
import os

import numpy as np
import pytest

from conftest import ROOT, case_params, make_depth_pair


def rand_rot(rng, scale=0.3):
    from scipy.spatial.transform import Rotation
    return Rotation.from_rotvec(rng.normal(size=3) * scale).as_matrix()


# ---------------------------------------------------------------------------------------- restated Eigen routines
def test_eigen3_against_numpy(oracle):
    rng = np.random.default_rng(1)
    for k in range(300):
        A = rng.normal(size=(3, 3)); A = A @ A.T
        if k % 3 == 0:   # flat patch like a wall: one tiny eigenvalue
            Q = rand_rot(rng, 1.0); A = Q @ np.diag([1e-5 * rng.random(), 0.5 + rng.random(), 0.6 + rng.random()]) @ Q.T
        A32 = A.astype(np.float32)
        ev, U = oracle.eigen3(A32)
        w, V = np.linalg.eigh(A32.astype(np.float64))
        assert np.all(np.diff(ev) >= -1e-6 * w[2])
        assert np.abs(ev - w).max() <= 2e-5 * w[2] + 1e-9
        # residual of each eigenpair
        for i in range(3):
            r = A32.astype(np.float64) @ U[:, i] - ev[i] * U[:, i]
            assert np.linalg.norm(r) <= 5e-5 * w[2]
        assert np.abs(U.T @ U - np.eye(3)).max() < 1e-4
    ev, U = oracle.eigen3(np.eye(3, dtype=np.float32) * 2)       # all eigenvalues equal -> identity eigenvectors
    assert np.allclose(ev, 2) and np.array_equal(U, np.eye(3, dtype=np.float32))


def test_ldlt_solve_against_numpy(oracle):
    rng = np.random.default_rng(2)
    for _ in range(200):
        J = rng.normal(size=(40, 6)) * rng.uniform(0.1, 30, size=6)
        H = (J.T @ J + 1001 * np.eye(6)).astype(np.float32)         # like aligner.cpp:92-94
        b = rng.normal(size=6).astype(np.float32) * 100
        x = oracle.ldlt_solve6(H, b)
        xr = np.linalg.solve(H.astype(np.float64), b.astype(np.float64))
        assert np.abs(x - xr).max() <= 2e-5 * np.abs(xr).max() + 1e-7
    x = oracle.ldlt_solve6(1001 * np.eye(6, dtype=np.float32), np.zeros(6, np.float32))   # no correspondences: dx = 0
    assert not x.any()


def test_se3_conventions_against_scipy(oracle):
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(3)
    for _ in range(100):
        q = rng.normal(size=4); q /= np.linalg.norm(q); q *= np.sign(q[3])
        t = rng.normal(size=3)
        v = np.concatenate([t, q[:3]]).astype(np.float32)
        T = oracle.v2t(v)
        R = Rotation.from_quat(q).as_matrix()
        # qw = sqrt(1 - |q|^2) in fp32 (bm_se3.h:14) loses bits for large rotations: 2e-5 is what fp32 gives
        assert np.abs(T[:3, :3] - R).max() < 2e-5 and np.abs(T[:3, 3] - t).max() < 1e-6 and np.array_equal(T[3], [0, 0, 0, 1])
        assert np.abs(oracle.t2v(T) - v).max() < 2e-5
    # mat2quat branches with non-positive trace (rotations by ~pi about each axis)
    for axis in range(3):
        rv = np.zeros(3); rv[axis] = 3.1
        R = Rotation.from_rotvec(rv).as_matrix().astype(np.float32)
        T = np.eye(4, dtype=np.float32); T[:3, :3] = R
        assert np.abs(oracle.v2t(oracle.t2v(T))[:3, :3] - R).max() < 2e-4


def test_projector_matrices_against_numpy(oracle):
    rng = np.random.default_rng(4)
    K = (525.0, 520.0, 319.5, 239.5)
    Km = np.array([[K[0], 0, K[2]], [0, K[1], K[3]], [0, 0, 1]])
    T = np.eye(4); T[:3, :3] = rand_rot(rng); T[:3, 3] = rng.normal(size=3)
    KRt, iKRt, iK = oracle.projector_matrices(K, T)
    Ti = np.linalg.inv(T)
    assert np.abs(KRt[:3, :3] - Km @ Ti[:3, :3]).max() < 1e-3 and np.abs(KRt[:3, 3] - Km @ Ti[:3, 3]).max() < 1e-3
    assert np.abs(iK - np.linalg.inv(Km)).max() < 1e-6
    assert np.abs(iKRt[:3, :3] - T[:3, :3] @ np.linalg.inv(Km)).max() < 1e-6 and np.abs(iKRt[:3, 3] - T[:3, 3]).max() < 1e-6


# ---------------------------------------------------------------------------------------- input conditioning
def test_depth_conversions(oracle):
    rng = np.random.default_rng(5)
    raw = rng.integers(0, 9000, size=(30, 40), dtype=np.uint16); raw[::3, ::4] = 0
    f = oracle.convert_16u_to_32f(raw)
    assert np.array_equal(f, np.where(raw > 0, np.float32(0.001) * raw.astype(np.float32), 0).astype(np.float32))
    g = f.copy(); g[0, 0] = np.finfo(np.float32).max
    u = oracle.convert_32f_to_16u(g)
    assert u[0, 0] == 0 and np.array_equal(u.ravel()[1:], (np.float32(1000.0) * g.ravel()[1:]).astype(np.uint16))
    # DepthImage_scale: step 1 is a copy; step 2 averages over the count of positive pixels, variance gate 0.01
    assert np.array_equal(oracle.depth_scale(f, 1), f)
    s2 = oracle.depth_scale(f, 2)
    blk = f[:2, 2:4].astype(np.float32)
    npos = (blk > 0).sum()
    if npos:
        mu = blk.sum(dtype=np.float32) / np.float32(npos)
        sig = (blk * blk).sum(dtype=np.float32) / np.float32(npos) - mu * mu
        assert s2[0, 1] == (0 if sig > 0.01 else mu)
    assert s2.shape == (15, 20)


# ---------------------------------------------------------------------------------------- projector
def test_unproject_project_roundtrip_and_ordering(oracle):
    rows, cols, K, conv, _ = case_params("small")
    depth, _, _, _, _ = make_depth_pair("small", 1)
    cp = oracle.converter_params(K=K, **conv)
    pts, idx = oracle.unproject(cp, depth)
    valid = (depth >= conv["min_distance"]) & (depth <= conv["max_distance"])
    assert len(pts) == valid.sum() and np.array_equal(idx >= 0, valid)
    assert np.array_equal(idx[valid], np.arange(valid.sum()))           # row-major rank (pinholepointprojector.cpp:125-127)
    r, c = np.nonzero(valid)
    assert np.allclose(pts[:, 2], depth[valid]) and np.allclose(pts[:, 0], (c - K[2]) / K[0] * depth[valid], atol=2e-5)
    assert np.all(pts[:, 3] == 1)
    # projecting back with the identity pose lands every point on its own pixel with its own depth
    pi, pd = oracle.project(K, np.eye(4), conv["min_distance"], conv["max_distance"], rows, cols, pts)
    assert np.array_equal(pi, idx)
    assert np.array_equal(pd[valid], depth[valid]) and np.all(pd[~valid] == np.finfo(np.float32).max)
    # intervals: int(max(fx, fy) * R / d), -1 where invalid (pinholepointprojector.h:264-274)
    itv = oracle.project_intervals(cp, depth)
    assert np.all(itv[~valid] == -1)
    exp = (np.float32(max(K[0], K[1])) * np.float32(conv["world_radius"]) * (np.float32(1.0) / depth[valid])).astype(np.int32)
    assert np.array_equal(itv[valid], exp)


def test_project_zbuffer_semantics(oracle):
    K = (30.0, 30.0, 15.5, 11.5)
    pts = np.array([[0, 0, 2, 1], [0, 0, 2, 1], [0, 0, 1.5, 1], [0, 0, 0.2, 1], [9, 0, 2, 1]], np.float32)
    idx, dep = oracle.project(K, np.eye(4), 0.5, 5.0, 24, 32, pts)
    # rounding half away from zero: 15.5 -> 16, 11.5 -> 12; nearest wins, ties keep the lowest index, too-near rejected
    assert idx[12, 16] == 2 and dep[12, 16] == np.float32(1.5)
    assert (idx >= 0).sum() == 1
    idx, _ = oracle.project(K, np.eye(4), 0.5, 5.0, 24, 32, pts[:2])
    assert idx[12, 16] == 0


# ---------------------------------------------------------------------------------------- integral image / stats
def test_integral_image_and_window_semantics(oracle):
    rows, cols, K, conv, _ = case_params("small")
    depth, _, _, _, _ = make_depth_pair("small", 2)
    cp = oracle.converter_params(K=K, **conv)
    pts, idx = oracle.unproject(cp, depth)
    I = oracle.integral_image(idx, pts)
    # independent float64 integral image
    v = np.zeros((10, rows, cols))
    m = idx >= 0
    p = pts[idx[m]].astype(np.float64)
    ch = [p[:, 0], p[:, 1], p[:, 2], np.ones(len(p)), p[:, 0] ** 2, p[:, 0] * p[:, 1], p[:, 0] * p[:, 2], p[:, 1] ** 2, p[:, 1] * p[:, 2], p[:, 2] ** 2]
    for k in range(10):
        v[k][m] = ch[k]
    ref = v.cumsum(2).cumsum(1)
    assert np.array_equal(I[3], ref[3].astype(np.float32))                 # the count channel is exact
    assert np.abs(I - ref).max() <= 2e-4 * np.abs(ref).max()                # fp32 prefix sums: ~1e-5 relative noise


def test_stats_window_is_the_reference_asymmetric_one(oracle):
    """getRegion (pointintegralimage.cpp:53-66): the window of pixel (r,c) with radius rad covers image
    x in (c-rad-1, c+rad-1], y in (r-rad-1, r+rad-1]: check Stats::n against brute-force counts."""
    rows, cols, K, conv, _ = case_params("small")
    depth, _, _, _, _ = make_depth_pair("small", 2)
    cp = oracle.converter_params(K=K, **conv)
    cloud, idx, itv = oracle.convert(cp, depth)
    a = cloud.arrays(stats=True)
    valid = (idx >= 0).astype(np.int64)
    S = np.zeros((rows + 1, cols + 1), np.int64); S[1:, 1:] = valid.cumsum(0).cumsum(1)
    rng = np.random.default_rng(0)
    checked = 0
    for _ in range(400):
        r, c = int(rng.integers(0, rows)), int(rng.integers(0, cols))
        if idx[r, c] < 0:
            continue
        rad = int(np.clip(itv[r, c], conv["min_image_radius"], conv["max_image_radius"]))
        cl = lambda x, hi: min(max(x, 0), hi)
        x0, x1 = cl(c - rad - 1, cols - 1), cl(c + rad - 1, cols - 1)
        y0, y1 = cl(r - rad - 1, rows - 1), cl(r + rad - 1, rows - 1)
        n = S[y1 + 1, x1 + 1] - S[y0 + 1, x1 + 1] - S[y1 + 1, x0 + 1] + S[y0 + 1, x0 + 1]     # x in (x0, x1], y in (y0, y1]
        got = a["npoints"][idx[r, c]]
        assert got == (n if n >= conv["min_points"] else 0), (r, c, rad, n, got)
        checked += 1
    assert checked > 100


def test_normals_match_the_scene_planes(oracle):
    """Back wall z = 4 has normal (0,0,-1) towards the camera; floor y = 1.2 has (0,-1,0)."""
    from g2o_frontend_amd import synth
    rows, cols, K, conv, _ = case_params("vga")
    mm = synth.render_depth_mm(11, np.eye(4), rows, cols, K, holes=0.03)
    depth = oracle.convert_16u_to_32f(mm)
    cp = oracle.converter_params(K=K, **conv)
    cloud, idx, _ = oracle.convert(cp, depth)
    a = cloud.arrays(stats=True)
    P, Nn = a["points"], a["normals"]
    # windows that straddle a depth discontinuity (wall vs sphere) are not planar patches: their largest
    # eigenvalue is ~0.1-1 m^2, a planar 0.1 m patch has ~3e-3
    ok = (np.abs(Nn[:, :3]).sum(1) > 0) & (a["eigenvalues"][:, 2] < 0.02)
    assert np.abs(np.linalg.norm(Nn[ok, :3], axis=1) - 1).max() < 1e-4 and np.all(Nn[:, 3] == 0)
    assert np.all((Nn[ok, :3] * P[ok, :3]).sum(1) <= 1e-6)                            # flipped to face the camera
    # interior of each plane only: a stats window that straddles a wall edge or touches a sphere mixes surfaces
    wall = ok & (np.abs(P[:, 2] - 4.0) < 0.004) & (np.abs(P[:, 0]) < 1.4) & (np.abs(P[:, 1]) < 0.7) & (a["curvature"] < 0.005)
    floor = ok & (np.abs(P[:, 1] - 1.2) < 0.004) & (np.abs(P[:, 0]) < 1.2) & (P[:, 2] < 3.3) & (a["curvature"] < 0.005)
    assert wall.sum() > 5000 and floor.sum() > 2000
    # 1 mm depth quantisation + fp32 prefix sums (pointintegralimage.cpp) leave a few 1e-2 of noise on the normals
    ew, ef = np.abs(Nn[wall, :3] - [0, 0, -1]).max(1), np.abs(Nn[floor, :3] - [0, -1, 0]).max(1)
    print(f"wall normals: median err {np.median(ew):.4f} q99 {np.quantile(ew, .99):.4f}; floor: median {np.median(ef):.4f} q99 {np.quantile(ef, .99):.4f}")
    assert np.median(ew) < 0.05 and np.quantile(ew, 0.99) < 0.25
    assert np.median(ef) < 0.05 and np.quantile(ef, 0.99) < 0.25
    # information matrices: flat points get U diag(1000,1,1) U^T -> trace 1002, largest eigen direction = normal
    flat = wall & (a["curvature"] < 0.02)
    om = a["omega_p"][flat].reshape(-1, 4, 4).transpose(0, 2, 1)[:, :3, :3]
    assert np.abs(np.trace(om, axis1=1, axis2=2) - 1002).max() < 0.5
    assert np.abs(np.einsum("ni,nij,nj->n", Nn[flat, :3], om, Nn[flat, :3]) - 1000).max() < 1.0
    assert np.all(a["omega_n"][flat].reshape(-1, 4, 4)[:, [0, 1, 2], [0, 1, 2]] == 100)


# ---------------------------------------------------------------------------------------- linearizer / aligner
@pytest.fixture(scope="module")
def small_clouds(oracle):
    rows, cols, K, conv, alig = case_params("small")
    ref, cur, Ttrue, _, _ = make_depth_pair("small", 1)
    cp = oracle.converter_params(K=K, **conv)
    cr, _, _ = oracle.convert(cp, ref); cc, _, _ = oracle.convert(cp, cur)
    return cr, cc, Ttrue


def test_linearizer_is_the_gradient_of_chi2(oracle, small_clouds):
    """b = sum J^T Omega e and H = sum J^T Omega J for e(dx) = v2t(dx) * T * (p_ref, n_ref) - (p_cur, n_cur):
    checked by central finite differences of chi2 over the same correspondence set (no robust kernel)."""
    cr, cc, _ = small_clouds
    rows, cols, K, conv, alig = case_params("small")
    ap = oracle.aligner_params(rows, cols, K=K, accumulate_fp64=1, **dict(alig, inlier_max_chi2=1e30))
    T = np.eye(4, dtype=np.float32)
    ri, _ = oracle.project(K, T, conv["min_distance"], conv["max_distance"], rows, cols, cr.arrays()["points"])
    ci, _ = oracle.project(K, T, conv["min_distance"], conv["max_distance"], rows, cols, cc.arrays()["points"])
    corr, _ = oracle.correspondences(ap, cr, cc, ri, ci, T)
    corr = corr[::7][:600]
    L0 = oracle.linearize(ap, cr, cc, corr, T)
    h = 1e-3
    grad = np.zeros(6)
    for k in range(6):
        dx = np.zeros(6, np.float32); dx[k] = h
        cp_ = oracle.linearize(ap, cr, cc, corr, oracle.v2t(dx) @ T)["chi2_fp64"]
        cm_ = oracle.linearize(ap, cr, cc, corr, oracle.v2t(-dx) @ T)["chi2_fp64"]
        grad[k] = (cp_ - cm_) / (2 * h)
    assert np.abs(grad - 2 * L0["b"]).max() <= 2e-3 * np.abs(L0["b"]).max()
    H = L0["H"].astype(np.float64)
    assert np.abs(H - H.T).max() <= 1e-4 * np.abs(H).max()
    assert np.linalg.eigvalsh(0.5 * (H + H.T)).min() > -1e-3 * np.abs(H).max()
    # Gauss-Newton step from H, b decreases chi2
    dx = np.linalg.solve(H + 1001 * np.eye(6), -L0["b"].astype(np.float64)).astype(np.float32)
    assert oracle.linearize(ap, cr, cc, corr, oracle.v2t(dx) @ T)["chi2_fp64"] < L0["chi2_fp64"]


def test_robust_kernel_and_inlier_counting(oracle, small_clouds):
    cr, cc, _ = small_clouds
    rows, cols, K, conv, alig = case_params("small")
    T = np.eye(4, dtype=np.float32)
    ri, _ = oracle.project(K, T, conv["min_distance"], conv["max_distance"], rows, cols, cr.arrays()["points"])
    ci, _ = oracle.project(K, T, conv["min_distance"], conv["max_distance"], rows, cols, cc.arrays()["points"])
    ap = oracle.aligner_params(rows, cols, K=K, **alig)
    corr, Kc = oracle.correspondences(ap, cr, cc, ri, ci, T)
    assert 0 < len(corr) <= Kc <= rows * cols
    thr = 20.0
    rob = oracle.linearize(oracle.aligner_params(rows, cols, K=K, **dict(alig, inlier_max_chi2=thr, robust_kernel=1)), cr, cc, corr, T)
    non = oracle.linearize(oracle.aligner_params(rows, cols, K=K, **dict(alig, inlier_max_chi2=thr, robust_kernel=0)), cr, cc, corr, T)
    big = oracle.linearize(oracle.aligner_params(rows, cols, K=K, **dict(alig, inlier_max_chi2=1e30)), cr, cc, corr, T)
    assert rob["inliers"] == len(corr) and non["inliers"] < len(corr) and big["inliers"] == len(corr)
    assert non["chi2"] < rob["chi2"] < big["chi2"]
    assert np.array_equal(rob["H"], big["H"])          # H is not scaled by the robust weight (linearizer.cpp:84-86)


def test_align_recovers_the_synthetic_motion(oracle, small_clouds):
    cr, cc, Ttrue = small_clouds
    rows, cols, K, conv, alig = case_params("small")
    ap = oracle.aligner_params(rows, cols, K=K, **alig)
    r = oracle.align(ap, cr, cc, images=True)
    assert np.abs(r["T"][:3, 3] - Ttrue[:3, 3]).max() < 5e-3 and np.abs(r["T"][:3, :3] - Ttrue[:3, :3]).max() < 5e-3
    chi = [it["chi2"] for it in r["iterations"]]
    assert chi[-1] < 0.1 * chi[0] and r["error"] == chi[-1] and r["inliers"] == r["iterations"][-1]["inliers"]
    assert np.abs(r["T"][:3, :3] @ r["T"][:3, :3].T - np.eye(3)).max() < 1e-5
    # fp64-accumulated mode agrees with the reference-faithful fp32 serial sums to the accuracy of those sums
    r64 = oracle.align(oracle.aligner_params(rows, cols, K=K, accumulate_fp64=1, **alig), cr, cc)
    assert abs(r64["iterations"][0]["chi2"] - chi[0]) <= 1e-4 * chi[0]
    assert np.abs(r64["T"] - r["T"]).max() < 1e-4
    # zero iterations: T = initial guess
    g = oracle.v2t(np.array([0.01, 0, 0, 0, 0.01, 0], np.float32))
    r0 = oracle.align(oracle.aligner_params(rows, cols, K=K, initial_guess=g, **dict(alig, outer_iterations=0)), cr, cc)
    assert np.array_equal(r0["T"], g)


def test_sensor_offset_consistency(oracle):
    """A cloud converted with sensor offset S equals the offset-free cloud moved by S (cloud.cpp:173-186),
    and aligning two such clouds recovers the same relative motion expressed in the offset frame."""
    from g2o_frontend_amd import synth
    rows, cols, K, conv, alig = case_params("small")
    ref, cur, Ttrue, _, _ = make_depth_pair("small", 3)
    S = synth.v2t(np.array([0.05, -0.02, 0.1, 0.02, -0.03, 0.04])).astype(np.float32)
    c0, _, _ = oracle.convert(oracle.converter_params(K=K, **conv), ref)
    c1, _, _ = oracle.convert(oracle.converter_params(K=K, sensor_offset=S, **conv), ref)
    a0, a1 = c0.arrays(), c1.arrays()
    assert np.abs(a1["points"][:, :3] - (a0["points"][:, :3] @ S[:3, :3].T + S[:3, 3])).max() < 1e-5
    assert np.abs(a1["normals"][:, :3] - a0["normals"][:, :3] @ S[:3, :3].T).max() < 1e-5
    assert np.array_equal(a0["curvature"], a1["curvature"])
    cpS = oracle.converter_params(K=K, sensor_offset=S, **conv)
    cr, _, _ = oracle.convert(cpS, ref); cc, _, _ = oracle.convert(cpS, cur)
    r = oracle.align(oracle.aligner_params(rows, cols, K=K, reference_sensor_offset=S, current_sensor_offset=S, **alig), cr, cc)
    expect = S.astype(np.float64) @ Ttrue @ np.linalg.inv(S.astype(np.float64))
    assert np.abs(r["T"] - expect).max() < 1e-2


def test_trig_mode_sensitivity(oracle):
    """Canonical mode (trig rounded once from double precision) vs literal float-libm calls as the reference makes them:
    quantifies how much of the converter output depends on the host libm's last bit."""
    rows, cols, K, conv, _ = case_params("small")
    depth, _, _, _, _ = make_depth_pair("small", 1)
    cp = oracle.converter_params(K=K, **conv)
    a = oracle.convert(cp, depth)[0].arrays(stats=True)
    oracle.set_trig_mode(True)
    try:
        b = oracle.convert(cp, depth)[0].arrays(stats=True)
    finally:
        oracle.set_trig_mode(False)
    assert np.array_equal(a["points"], b["points"]) and np.array_equal(a["npoints"], b["npoints"])
    va, vb = np.abs(a["normals"]).sum(1) > 0, np.abs(b["normals"]).sum(1) > 0
    assert (va != vb).sum() <= 2
    both = va & vb
    dn = np.abs(a["normals"][both] - b["normals"][both]).max(1)
    same = float((dn == 0).mean())
    print(f"trig modes: {same * 100:.1f}% of normals bit-identical, median |dn| {np.median(dn):.2e}, max {dn.max():.2e}")
    assert np.quantile(dn, 0.99) < 1e-4


def test_canonical_trig_is_the_correctly_rounded_value(oracle):
    """The eigensolver's trig in canonical mode (fixed double-precision + - * / algorithms, the very operations the kernels execute) against
    the double libm rounded to float, over 2 M random arguments of all magnitudes and the special points (zeros of both signs, the axes, the
    diagonal): equal everywhere -- an error below 1e-15 changes the rounded float about once in 1e8 calls, so a couple of differing values
    are tolerated, and none may differ by more than one ulp."""
    rng = np.random.default_rng(7)
    n = 2_000_000
    y = np.abs(rng.standard_normal(n)).astype(np.float32) * np.float32(10.0) ** rng.integers(-7, 3, n).astype(np.float32)
    x = rng.standard_normal(n).astype(np.float32) * np.float32(10.0) ** rng.integers(-7, 3, n).astype(np.float32)
    y[:1000] = 0; x[:500] = np.float32(-0.0); x[500:1000] = 0.0; y[1000:2000] = 1.0; x[1000:1500] = 1.0; x[1500:2000] = -1.0
    x[2000:2500] = 0.0; x[2500:3000] = np.float32(-0.0)
    a, b = oracle.trig_eval(0, y, x), oracle.trig_eval(2, y, x)
    for got, want, name in zip(a, b, ("theta", "cos", "sin")):
        diff = got.view(np.int32).astype(np.int64) - want.view(np.int32).astype(np.int64)
        assert int((diff != 0).sum()) <= 3 and int(np.abs(diff).max()) <= 1, (name, int((diff != 0).sum()))
    th = a[0]
    assert th.min() >= 0 and th.max() <= np.float32(np.pi / 3) * (1 + 1e-6)
    assert np.all(a[0][:500] == np.float32(np.pi) * np.float32(1.0 / 3.0)) and np.all(a[0][500:1000] == 0)      # atan2(+0, -0) = pi, atan2(+0, +0) = 0


# ---------------------------------------------------------------------------------------- generator
def test_synthetic_generator_is_deterministic_and_plausible():
    from g2o_frontend_amd import synth
    a, b, T = synth.make_pair(5, 120, 160, synth.scaled_K(synth.K_VGA, 4))
    a2, b2, T2 = synth.make_pair(5, 120, 160, synth.scaled_K(synth.K_VGA, 4))
    assert np.array_equal(a, a2) and np.array_equal(b, b2) and np.array_equal(T, T2)
    assert a.dtype == np.uint16 and 0.02 < (a == 0).mean() < 0.04
    z = a[a > 0]
    assert 700 < z.min() and z.max() <= 4000
    assert np.abs(T[:3, 3]).max() <= 0.05 and np.abs(T[:3, :3] - np.eye(3)).max() < 0.1
    assert not np.array_equal(a, synth.make_pair(6, 120, 160, synth.scaled_K(synth.K_VGA, 4))[0])
    tr = synth.trajectory(1, 20)
    for p, q in zip(tr[:-1], tr[1:]):
        d = np.linalg.inv(p) @ q
        assert np.linalg.norm(d[:3, 3]) <= 0.021 and np.arccos(np.clip((np.trace(d[:3, :3]) - 1) / 2, -1, 1)) <= np.deg2rad(1.01)


def test_parallel_align_variant_keeps_the_canonical_list(oracle):
    """The timed CPU baseline's OpenMP variant of CorrespondenceFinder::compute / Linearizer::update (the reference's thread partition
    without its remainder dropping: correspondencefinder.cpp:38-51, linearizer.cpp:32-39): the correspondence list is the one-thread list
    exactly, whatever the thread count; H, b, chi2 differ only by the order of the per-thread partial sums."""
    O = oracle
    rows, cols, K, conv, alig = case_params("small")
    ref, cur, _, _, _ = make_depth_pair("small", 1)
    cp = O.converter_params(K=K, **conv); ap = O.aligner_params(rows, cols, K=K, **alig)
    cr, ridx, _ = O.convert(cp, ref); cc, cidx, _ = O.convert(cp, cur)
    T = np.eye(4, dtype=np.float32)
    try:
        O.set_parallel_align(False)
        c1, k1 = O.correspondences(ap, cr, cc, ridx, cidx, T)
        l1 = O.linearize(ap, cr, cc, c1, T)
        for threads in (3, 7):                                   # 120 rows: 7 does not divide them -- the reference would drop a row
            O.set_num_threads(threads); O.set_parallel_align(True)
            cN, kN = O.correspondences(ap, cr, cc, ridx, cidx, T)
            assert kN == k1 and np.array_equal(cN, c1)
            lN = O.linearize(ap, cr, cc, c1, T)
            assert lN["inliers"] == l1["inliers"]
            assert abs(lN["chi2"] - l1["chi2"]) <= 1e-4 * abs(l1["chi2"])
            assert np.abs(lN["H"] - l1["H"]).max() <= 1e-4 * np.abs(l1["H"]).max()
    finally:
        O.set_parallel_align(False); O.set_num_threads(8)


def test_fast_build_of_the_oracle_is_bit_identical():
    """bench.py times the -O3 -march=native build of the oracle source; same results as the -O2 checker build."""
    import json
    import subprocess
    import sys
    code = (
        "import os, sys, json, hashlib\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "if sys.argv[1] == 'fast': os.environ['PWN_ORACLE_VARIANT'] = 'fast'\n"
        "from conftest import case_params, make_depth_pair\n"
        "from oracle import oracle as O\n"
        "rows, cols, K, conv, alig = case_params('small')\n"
        "ref, cur, _, _, _ = make_depth_pair('small', 1)\n"
        "cp = O.converter_params(K=K, **conv); ap = O.aligner_params(rows, cols, K=K, **alig)\n"
        "cr, _, _ = O.convert(cp, ref); cc, _, _ = O.convert(cp, cur)\n"
        "r = O.align(ap, cr, cc)\n"
        "a = cr.arrays()\n"
        "h = hashlib.sha1(b''.join(a[k].tobytes() for k in ('points', 'normals', 'curvature', 'omega_p', 'omega_n'))).hexdigest()\n"
        "print(json.dumps(dict(h=h, chi2=[float(i['chi2']) for i in r['iterations']], T=r['T'].astype(float).tolist())))\n"
    ) % (ROOT, os.path.join(ROOT, "tests"))
    outs = []
    for variant in ("canonical", "fast"):
        p = subprocess.run([sys.executable, "-c", code, variant], capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        outs.append(json.loads(p.stdout.strip().splitlines()[-1]))
    assert outs[0] == outs[1]
