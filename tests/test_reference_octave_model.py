"""The one executable model of this path that the reference repository itself holds: the Octave scripts of the PWN least squares
(g2o_frontend/octave/pwn/{v2t,t2v,quat2mat,mat2quat,pwn_remapPoint,pwn_jacobian,pwn_iteration}.m, g2o_frontend/octave/skew.m; driven by
octave/PWNTest.m, which prints and asserts nothing).  Octave is not installed here, so the formulas are restated below in numpy, function by
function (file:line cited), and the ORACLE is checked against them -- the SE(3) chart of bm_se3.h and the linearizer's H, b, chi2.  This pins
the oracle's conventions (vector part of the quaternion as rotation increment, the factor 2 in skew(), the sign and block layout of the
Jacobian) on an artefact of the reference; it does not pin the converter or the projector, for which the reference holds no model (DESIGN.md
section 2: parity unpinned)."""
import numpy as np
import pytest


# ---- octave/pwn/quat2mat.m:5-23, v2t.m:4-8, mat2quat.m:5-10, t2v.m:5-9, octave/skew.m:6-13, pwn_remapPoint.m:6-10, pwn_jacobian.m:7-12
def m_quat2mat(q):
    qx, qy, qz = q
    qw = np.sqrt(1.0 - q @ q) if q @ q <= 1 else 0.0
    if q @ q > 1:
        qx = qy = 0.0
    return np.array([[qw * qw + qx * qx - qy * qy - qz * qz, 2 * (qx * qy - qw * qz), 2 * (qx * qz + qw * qy)],
                     [2 * (qx * qy + qz * qw), qw * qw - qx * qx + qy * qy - qz * qz, 2 * (qy * qz - qx * qw)],
                     [2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), qw * qw - qx * qx - qy * qy + qz * qz]])


def m_v2t(x):
    X = np.eye(4); X[:3, 3] = x[:3]; X[:3, :3] = m_quat2mat(np.asarray(x[3:6], float)); return X


def m_mat2quat(R):
    qw4 = 2 * np.sqrt(1 + R[0, 0] + R[1, 1] + R[2, 2])
    return np.array([(R[2, 1] - R[1, 2]) / qw4, (R[0, 2] - R[2, 0]) / qw4, (R[1, 0] - R[0, 1]) / qw4])


def m_t2v(X):
    return np.concatenate([X[:3, 3], m_mat2quat(X[:3, :3])])


def m_skew(t):
    tx, ty, tz = t
    return np.array([[0, -tz, ty], [tz, 0, -tx], [-ty, tx, 0]])


def m_remap(X, p):
    return np.concatenate([X[:3, :3] @ p[:3] + X[:3, 3], X[:3, :3] @ p[3:6]])


def m_jacobian(X, p):
    J = np.zeros((6, 6))
    J[:3, :3] = X[:3, :3]
    J[:3, 3:] = -X[:3, :3] @ (2 * m_skew(p[:3]))
    J[3:, 3:] = -X[:3, :3] @ (2 * m_skew(p[3:6]))
    return J


def m_iteration_sums(Pi, Pj, Omega, X):
    """the accumulation loop of pwn_iteration.m:12-27 (H, b, err) without its solve"""
    b = np.zeros(6); H = np.zeros((6, 6)); err = 0.0
    for i in range(Pi.shape[1]):
        e = Pi[:, i] - m_remap(X, Pj[:, i])
        J = -m_jacobian(X, Pj[:, i])
        b += J.T @ Omega @ e
        H += J.T @ Omega @ J
        err += e @ Omega @ e
    return H, b, err


def _clouds(oracle, rng, n, omega_p, omega_n, X):
    """reference cloud = random points with unit normals; current cloud = the reference moved by X plus a little noise"""
    ref = np.zeros((n, 6)); ref[:, :3] = rng.uniform(-1, 1, (n, 3)) + (0, 0, 2.5)
    nr = rng.normal(size=(n, 3)); ref[:, 3:] = nr / np.linalg.norm(nr, axis=1, keepdims=True)
    cur = np.array([m_remap(X, p) for p in ref]) + rng.normal(scale=2e-3, size=(n, 6))
    cur[:, 3:] /= np.linalg.norm(cur[:, 3:], axis=1, keepdims=True)

    def cloud(a):
        P = np.ones((n, 4), np.float32); P[:, :3] = a[:, :3]
        N = np.zeros((n, 4), np.float32); N[:, :3] = a[:, 3:]
        op = np.zeros((n, 4, 4), np.float32); op[:, :3, :3] = omega_p
        on = np.zeros((n, 4, 4), np.float32); on[:, :3, :3] = omega_n
        # column-major 4x4 per point
        return oracle.Cloud.from_arrays(P, N, np.full(n, 0.01, np.float32), op.transpose(0, 2, 1).reshape(n, 16), on.transpose(0, 2, 1).reshape(n, 16))
    return ref, cur, cloud(ref), cloud(cur)


def test_se3_chart_matches_the_octave_model(oracle):
    rng = np.random.default_rng(5)
    for _ in range(200):
        x = np.concatenate([rng.uniform(-2, 2, 3), rng.uniform(-0.5, 0.5, 3)])
        X = m_v2t(x)
        assert np.abs(oracle.v2t(x.astype(np.float32)) - X).max() < 2e-6                      # bm_se3.h:37-43 vs v2t.m / quat2mat.m
        assert np.abs(oracle.t2v(X.astype(np.float32)) - m_t2v(X)).max() < 2e-6               # bm_se3.h:45-52 vs t2v.m / mat2quat.m (w > 0)
        assert np.abs(m_t2v(m_v2t(x)) - x).max() < 1e-12                                     # the model's own round trip


def test_linearizer_matches_the_octave_model(oracle):
    """H, b, chi2 of Linearizer::update (linearizer.cpp:17-115, restated in the oracle) against pwn_iteration.m's sums with one information
    matrix for all points.  Mapping: the model moves Pj onto Pi (e = pi - X pj); the C++ moves the reference cloud onto the current one
    with invT (e = invT p_ref - p_cur): Pi = current, Pj = reference, X = invT gives e_model = -e_cpp and, at X = I, J_model = -J_cpp (the
    C++ Jacobian [I, skew(p'); 0, skew(n')] with skew(v) = -2 [v]x, bm_se3.h:54-66, is pwn_jacobian.m's J), so H and b coincide; chi2
    coincides for every X."""
    rng = np.random.default_rng(6)
    omega_p = np.diag([1000.0, 1.0, 1.0]); Q = np.linalg.qr(rng.normal(size=(3, 3)))[0]; omega_p = Q @ omega_p @ Q.T      # a full symmetric 3x3
    omega_n = np.eye(3) * 100.0
    Omega = np.zeros((6, 6)); Omega[:3, :3] = omega_p; Omega[3:, 3:] = omega_n
    n = 400
    corr = np.stack([np.arange(n), np.arange(n)], 1).astype(np.int32)
    ap = oracle.aligner_params(120, 160, accumulate_fp64=1, **dict(oracle.QVGA4_CONF_ALIGNER, inlier_max_chi2=1e30))
    # (1) at the identity: H, b, chi2
    X = m_v2t(np.array([0.004, -0.003, 0.002, 0.001, -0.002, 0.0015]))       # the two clouds differ by this small motion + noise
    ref, cur, cref, ccur = _clouds(oracle, rng, n, omega_p, omega_n, X)
    H, b, err = m_iteration_sums(cur.T, ref.T, Omega, np.eye(4))
    o = oracle.linearize(ap, cref, ccur, corr, np.eye(4, dtype=np.float32))
    assert o["inliers"] == n
    assert abs(o["chi2_fp64"] - err) <= 2e-5 * err
    assert np.abs(o["H"] - H).max() <= 2e-5 * np.abs(H).max()
    assert np.abs(o["b"] - b).max() <= 2e-5 * np.abs(b).max() + 1e-6 * np.abs(H).max()
    # the model's step from the identity moves the reference cloud onto the current one: dx = -H \\ b, Xnew = v2t(dx)  (pwn_iteration.m:28-30)
    Xnew = m_v2t(-np.linalg.solve(H, b))
    assert np.abs(Xnew - X).max() < 2e-3
    # (2) chi2 at a general transform
    for _ in range(5):
        Xg = m_v2t(np.concatenate([rng.uniform(-0.05, 0.05, 3), rng.uniform(-0.02, 0.02, 3)]))
        _, _, err = m_iteration_sums(cur.T, ref.T, Omega, Xg)
        o = oracle.linearize(ap, cref, ccur, corr, Xg.astype(np.float32))
        assert abs(o["chi2_fp64"] - err) <= 5e-5 * err


def test_PWNTest_scenario_chi2_along_the_models_own_trajectory(oracle):
    """octave/PWNTest.m's scenario at its first noise level (100 random points with unit normals in a 100 m cube, ground truth transform
    v2t([100 200 300 .5 .5 .5]) = 120 degrees about (1,1,1), Omega = diag(1,1,1,100,100,100), start at the identity), run with the numpy
    restatement of pwn_solve.m / pwn_iteration.m (update Xnew = X * v2t(-H \\ b)).  The model must find the ground truth (its script prints the
    transform error and asserts nothing), and the ORACLE's chi2 (Linearizer::update, linearizer.cpp:17-115) must agree with the model's at
    every iterate of that trajectory -- large rotations and 100 m coordinates, where the small-motion cases of the test above do not go."""
    rng = np.random.default_rng(11)
    n, tscale = 100, 100.0
    Pi = rng.uniform(-0.5, 0.5, (6, n)); Pi[:3] *= tscale; Pi[3:] /= np.linalg.norm(Pi[3:], axis=0, keepdims=True)       # PWNTest.m:7-11
    gtX = m_v2t(np.array([100.0, 200.0, 300.0, 0.5, 0.5, 0.5]))                                                          # :14-15
    Pj = np.stack([m_remap(gtX, Pi[:, i]) for i in range(n)], 1)                                                         # :23 (noise level 1: none)
    Omega = np.eye(6); Omega[3:, 3:] *= 100.0                                                                           # :25-28
    # the oracle's clouds: Pi = current (the fixed side of e = pi - X pj), Pj = reference (moved by invT = X), one information matrix for all
    def cloud(P6):
        P = np.ones((n, 4), np.float32); P[:, :3] = P6[:3].T
        N = np.zeros((n, 4), np.float32); N[:, :3] = P6[3:].T
        op = np.zeros((n, 4, 4), np.float32); op[:, :3, :3] = Omega[:3, :3]
        on = np.zeros((n, 4, 4), np.float32); on[:, :3, :3] = Omega[3:, 3:]
        return oracle.Cloud.from_arrays(P, N, np.full(n, 0.01, np.float32), op.transpose(0, 2, 1).reshape(n, 16), on.transpose(0, 2, 1).reshape(n, 16))
    ccur, cref = cloud(Pi), cloud(Pj)
    corr = np.stack([np.arange(n), np.arange(n)], 1).astype(np.int32)
    ap = oracle.aligner_params(120, 160, accumulate_fp64=1, **dict(oracle.QVGA4_CONF_ALIGNER, inlier_max_chi2=1e30))
    X = np.eye(4); errs = []
    for it in range(40):                                                                                                # :47-52, pwn_solve.m:13-20
        H, b, err = m_iteration_sums(Pi, Pj, Omega, X)
        o = oracle.linearize(ap, cref, ccur, corr, X.astype(np.float32))
        assert o["inliers"] == n
        assert abs(o["chi2_fp64"] - err) <= 2e-4 * err + 1e-3, (it, o["chi2_fp64"], err)      # fp32 terms of 100 m coordinates against float64
        errs.append(err)
        X = X @ m_v2t(-np.linalg.solve(H, b))                                                                           # pwn_iteration.m:28-30
    assert errs[0] > 1e6 and errs[-1] < 1e-12 * errs[0]                                                                 # the model converges ...
    assert np.abs(m_t2v(X @ gtX)).max() < 1e-6                                                                          # ... to the ground truth (PWNTest.m:57-58)


def test_products_se3_chart_against_the_octave_model_directly():
    """pwn_hip_v2t / pwn_hip_t2v / pwn_hip_iso_mul / pwn_hip_iso_inverse (host compilations of the device functions k_solve_update runs;
    no GPU needed) against v2t.m / t2v.m / quat2mat.m / mat2quat.m -- the product's SE(3) chart on the reference's model, without the oracle."""
    from g2o_frontend_amd import api
    rng = np.random.default_rng(8)
    for _ in range(200):
        x = np.concatenate([rng.uniform(-2, 2, 3), rng.uniform(-0.5, 0.5, 3)])
        X = m_v2t(x)
        assert np.abs(api.v2t(x.astype(np.float32)) - X).max() < 2e-6
        assert np.abs(api.t2v(X.astype(np.float32)) - m_t2v(X)).max() < 2e-6
        Y = m_v2t(np.concatenate([rng.uniform(-1, 1, 3), rng.uniform(-0.4, 0.4, 3)]))
        assert np.abs(api.iso_mul(X.astype(np.float32), Y.astype(np.float32)) - X @ Y).max() < 5e-6
        assert np.abs(api.iso_inverse(X.astype(np.float32)) - np.linalg.inv(X)).max() < 5e-6
