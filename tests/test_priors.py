"""SURVEY.md §8(f) row 2 (second half): SE(3) priors in Aligner::align (pwn_core/aligner.cpp:34-47,96-108, se3_prior.{h,cpp})."""
import numpy as np
import pytest

from conftest import case_params, make_depth_pair


@pytest.fixture()
def small(oracle):
    rows, cols, K, conv, alig = case_params("small")
    ref, cur, Ttrue, _, _ = make_depth_pair("small", 1)
    cp = oracle.converter_params(K=K, **conv)
    cr, _, _ = oracle.convert(cp, ref); cc, _, _ = oracle.convert(cp, cur)
    yield dict(rows=rows, cols=cols, K=K, conv=conv, alig=alig, ref=ref, cur=cur, Ttrue=Ttrue, cr=cr, cc=cc)
    oracle.clear_priors()


def test_prior_pulls_the_solution_cpu(oracle, small):
    """A strong relative prior centred on a wrong transform drags the estimate towards it; a prior on the true motion does not move it."""
    from g2o_frontend_amd import synth
    d = small
    ap = oracle.aligner_params(d["rows"], d["cols"], K=d["K"], accumulate_fp64=1, **d["alig"])
    free = oracle.align(ap, d["cr"], d["cc"])
    wrong = synth.v2t(np.array([0.2, 0.0, 0.0, 0.0, 0.0, 0.0])).astype(np.float32)
    oracle.clear_priors(); oracle.add_prior(0, wrong, np.eye(6) * 1e7)
    pulled = oracle.align(ap, d["cr"], d["cc"])
    assert np.linalg.norm(pulled["T"][:3, 3] - wrong[:3, 3]) < np.linalg.norm(free["T"][:3, 3] - wrong[:3, 3]) * 0.5
    oracle.clear_priors(); oracle.add_prior(0, free["T"], np.eye(6) * 1e5)
    same = oracle.align(ap, d["cr"], d["cc"])
    assert np.abs(same["T"] - free["T"]).max() < 2e-3
    # absolute prior with reference R and mean M acts like a relative prior with mean R^-1 M
    R = synth.v2t(np.array([0.05, -0.02, 0.01, 0.01, 0.0, -0.01])).astype(np.float32)
    M = (R.astype(np.float64) @ wrong.astype(np.float64)).astype(np.float32)
    oracle.clear_priors(); oracle.add_prior(1, M, np.eye(6) * 1e7, reference_transform=R)
    pulled_abs = oracle.align(ap, d["cr"], d["cc"])
    assert np.abs(pulled_abs["T"] - pulled["T"]).max() < 5e-3


@pytest.mark.gpu
@pytest.mark.parametrize("kind", [0, 1])
def test_align_with_priors_matches_oracle(oracle, small, kind):
    from g2o_frontend_amd import api, synth
    from test_gpu_parity import gpu_objects, _check_alignment
    d = small
    ctx = api.Context(0, d["rows"], d["cols"], 2)
    _, converter, aligner = gpu_objects(ctx, "small")
    gref, gcur = api.Cloud(ctx, d["rows"] * d["cols"]), api.Cloud(ctx, d["rows"] * d["cols"])
    converter.compute(gref, d["ref"]); converter.compute(gcur, d["cur"])
    aligner.setReferenceCloud(gref); aligner.setCurrentCloud(gcur)
    mean = synth.v2t(np.array([0.06, -0.03, -0.02, 0.01, -0.015, 0.01])).astype(np.float32)
    reft = synth.v2t(np.array([0.02, 0.01, -0.01, 0.0, 0.01, 0.0])).astype(np.float32)
    info = (np.diag([4e5, 4e5, 4e5, 2e6, 2e6, 2e6]) + 1e4).astype(np.float32)
    ap = oracle.aligner_params(d["rows"], d["cols"], K=d["K"], accumulate_fp64=1, **d["alig"])
    oracle.clear_priors()
    free = oracle.align(ap, d["cr"], d["cc"])
    if kind == 0:
        oracle.add_prior(0, mean, info); aligner.addRelativePrior(mean, info)
    else:
        oracle.add_prior(1, mean, info, reference_transform=reft); aligner.addAbsolutePrior(reft, mean, info)
    o = oracle.align(ap, d["cr"], d["cc"])
    g = aligner.align()
    assert np.abs(o["T"] - free["T"]).max() > 1e-3            # the prior matters in this set-up
    _check_alignment(o, g)
    # priors are cleared by setting a cloud (aligner.h:60-63): back to the prior-free result
    aligner.setCurrentCloud(gcur)
    g2 = aligner.align()
    assert np.abs(g2["T"] - free["T"]).max() < 1e-5
    ctx.close()
