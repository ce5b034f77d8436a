"""SURVEY.md §8(f) row 2 (first half): Aligner::_computeStatistics (pwn_core/aligner.cpp:152-199, unscented.h:23-65)."""
import ctypes as C

import numpy as np
import pytest

from conftest import case_params, make_depth_pair


def stats_float64(H, T):
    """Independent float64 model of aligner.cpp:152-199 (numpy inverse / cholesky / svd, scipy-free SE(3) helpers)."""
    from g2o_frontend_amd import synth
    n = 6
    sigma = np.linalg.inv(H.astype(np.float64) + np.eye(n))
    alpha, beta = 1e-3, 2.0
    lam = alpha * alpha * n
    wi = 1.0 / (2 * (n + lam))
    L = np.linalg.cholesky(sigma * (n + lam))
    pts = [np.zeros(n)] + [s * L[:, i] for i in range(n) for s in (1.0, -1.0)]
    w_i = [lam / (n + lam)] + [wi] * (2 * n); w_p = [lam / (n + lam) + (1 - alpha * alpha + beta)] + [wi] * (2 * n)

    def t2v(X):
        R = X[:3, :3]
        qw = np.sqrt(max(0.0, 1 + np.trace(R))) / 2
        q = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]) / (4 * qw)
        return np.concatenate([X[:3, 3], q])
    rem = [t2v(T.astype(np.float64) @ np.linalg.inv(synth.v2t(p))) for p in pts]
    mean = sum(w * p for w, p in zip(w_i, rem))
    cov = sum(w * np.outer(p - mean, p - mean) for w, p in zip(w_p, rem))
    om = np.linalg.inv(cov)
    sv_t = np.linalg.svd(om[:3, :3], compute_uv=False); sv_r = np.linalg.svd(om[3:, 3:], compute_uv=False)
    return mean, om, sv_t[0] / sv_t[2], sv_r[0] / sv_r[2]


def random_system(rng):
    from g2o_frontend_amd import synth
    J = rng.normal(size=(200, 6)) * np.array([30, 30, 30, 60, 60, 60])
    H = (J.T @ J).astype(np.float32)
    T = synth.v2t(np.concatenate([rng.uniform(-0.2, 0.2, 3), rng.uniform(-0.05, 0.05, 3)])).astype(np.float32)
    return H, T


def test_oracle_statistics_against_float64_model(oracle):
    rng = np.random.default_rng(0)
    for _ in range(20):
        H, T = random_system(rng)
        s = oracle.compute_statistics(H, T)
        mean, om, tr, rr = stats_float64(H, T)
        assert np.abs(s["mean"] - mean).max() < 1e-5
        assert np.abs(s["omega"] - om).max() <= 2e-2 * np.abs(om).max()          # fp32 6x6 inverse of an fp32 covariance
        assert abs(s["translationalEigenRatio"] - tr) <= 2e-2 * tr and abs(s["rotationalEigenRatio"] - rr) <= 2e-2 * rr
        assert np.abs(s["mean"] - oracle.t2v(T)).max() < 1e-4                       # the remapped mean is the solution itself


def test_product_host_statistics_match_oracle(oracle):
    """pwn_hip_compute_statistics is host code of the product (no GPU needed)."""
    from g2o_frontend_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(1)
    for _ in range(20):
        H, T = random_system(rng)
        o = oracle.compute_statistics(H, T)
        Hc = np.ascontiguousarray(H.T.reshape(-1)); Tc = np.ascontiguousarray(T.T.reshape(-1))
        mean = np.empty(6, np.float32); om = np.empty(36, np.float32); tr, rr = C.c_float(0), C.c_float(0)
        L.pwn_hip_compute_statistics(Hc.ctypes.data_as(C.c_void_p), Tc.ctypes.data_as(C.c_void_p), mean.ctypes.data_as(C.c_void_p),
                                     om.ctypes.data_as(C.c_void_p), C.byref(tr), C.byref(rr))
        om = om.reshape(6, 6).T
        assert np.abs(mean - o["mean"]).max() < 1e-6
        assert np.abs(om - o["omega"]).max() <= 1e-4 * np.abs(o["omega"]).max()
        assert abs(tr.value - o["translationalEigenRatio"]) <= 1e-4 * tr.value and abs(rr.value - o["rotationalEigenRatio"]) <= 1e-4 * rr.value


@pytest.mark.gpu
@pytest.mark.parametrize("name,seed", [("small", 1), ("vga", 0)])
def test_align_statistics_match_oracle(oracle, name, seed):
    from g2o_frontend_amd import api
    from test_gpu_parity import gpu_objects, oracle_params
    rows, cols, K, conv, _ = case_params(name)
    ref, cur, Ttrue, _, _ = make_depth_pair(name, seed)
    cp, ap = oracle_params(oracle, name, accumulate_fp64=1)
    oref, _, _ = oracle.convert(cp, ref); ocur, _, _ = oracle.convert(cp, cur)
    o = oracle.align(ap, oref, ocur)
    os_ = oracle.align_statistics(ap, oref, ocur, o["T"])
    ctx = api.Context(0, rows, cols, 2)
    _, converter, aligner = gpu_objects(ctx, name)
    gref, gcur = api.Cloud(ctx, rows * cols), api.Cloud(ctx, rows * cols)
    converter.compute(gref, ref); converter.compute(gcur, cur)
    aligner.setReferenceCloud(gref); aligner.setCurrentCloud(gcur)
    plain = aligner.align()
    g = aligner.align(statistics=True)
    assert np.array_equal(plain["T"], g["T"]) and np.array_equal(plain["chi2"], g["chi2"])      # statistics do not disturb the alignment
    st = aligner._statistics
    hs = np.abs(os_["H"]).max()
    tol = 1e-4 if name == "vga" else 2e-2          # free-running: one flipped correspondence at 120x160 is ~1e-4..1e-2 of H
    assert np.abs(st["H"] - os_["H"]).max() <= tol * hs
    assert np.abs(aligner.omega() - os_["omega"]).max() <= 10 * tol * np.abs(os_["omega"]).max()
    assert abs(aligner.translationalEigenRatio() - os_["translationalEigenRatio"]) <= 10 * tol * os_["translationalEigenRatio"]
    assert abs(aligner.rotationalEigenRatio() - os_["rotationalEigenRatio"]) <= 10 * tol * os_["rotationalEigenRatio"]
    assert np.abs(st["mean"] - oracle.t2v(g["T"])).max() < 1e-4
    assert aligner.solutionValid() == (not (os_["rotationalEigenRatio"] > 50 or os_["translationalEigenRatio"] > 50))
    # host statistics of the GPU's own H are what the host function gives for that H (exactly the same code path)
    again = oracle.compute_statistics(st["H"], g["T"])
    assert np.abs(again["omega"] - aligner.omega()).max() <= 1e-3 * np.abs(again["omega"]).max()
    ctx.close()
