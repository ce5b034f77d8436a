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


@pytest.mark.gpu
def test_priors_and_statistics_together(oracle, small):
    """The reference adds the priors inside the loop (aligner.cpp:96-108) AND runs _computeStatistics afterwards (:127) in the same
    align(); neither may be dropped when both are asked for (the two mirrors each dropped one of them)."""
    from g2o_frontend_amd import api, synth
    from test_gpu_parity import gpu_objects, _check_alignment
    d = small
    ctx = api.Context(0, d["rows"], d["cols"], 2)
    _, converter, aligner = gpu_objects(ctx, "small")
    gref, gcur = api.Cloud(ctx, d["rows"] * d["cols"]), api.Cloud(ctx, d["rows"] * d["cols"])
    converter.compute(gref, d["ref"]); converter.compute(gcur, d["cur"])
    mean = synth.v2t(np.array([0.06, -0.03, -0.02, 0.01, -0.015, 0.01])).astype(np.float32)
    info = (np.diag([4e5, 4e5, 4e5, 2e6, 2e6, 2e6]) + 1e4).astype(np.float32)
    ap = oracle.aligner_params(d["rows"], d["cols"], K=d["K"], accumulate_fp64=1, **d["alig"])
    oracle.clear_priors(); oracle.add_prior(0, mean, info)
    o = oracle.align(ap, d["cr"], d["cc"])
    os_ = oracle.align_statistics(ap, d["cr"], d["cc"], o["T"])      # the 11th update: H of the linearizer alone, no prior terms (:165-170)
    aligner.setReferenceCloud(gref); aligner.setCurrentCloud(gcur)
    aligner.addRelativePrior(mean, info)
    only_priors = aligner.align()
    aligner._omega[:] = -1.0
    both = aligner.align(statistics=True)
    assert np.array_equal(only_priors["T"], both["T"]) and np.array_equal(only_priors["chi2"], both["chi2"])      # the priors were not dropped
    _check_alignment(o, both)
    st = aligner._statistics
    hs = np.abs(os_["H"]).max()
    assert np.abs(st["H"] - os_["H"]).max() <= 2e-2 * hs                          # free-running 120x160 bar of test_statistics
    assert np.abs(aligner.omega() - os_["omega"]).max() <= 0.2 * np.abs(os_["omega"]).max()
    again = oracle.compute_statistics(st["H"], both["T"])                          # the statistics are those of the returned H and T
    assert np.abs(again["omega"] - aligner.omega()).max() <= 1e-3 * np.abs(again["omega"]).max()
    assert np.abs(st["mean"] - oracle.t2v(both["T"])).max() < 1e-4
    ctx.close()


@pytest.mark.gpu
def test_fixed_tag_paths_do_not_poison_later_alignments(oracle, small):
    """z-buffer epoch tags: a smaller tag wins atomicMin.  A priors alignment with many outer iterations between two one-iteration
    alignments must not leave words behind that beat the later call's tags: same result as on a fresh context."""
    from g2o_frontend_amd import api, synth
    from test_gpu_parity import gpu_objects
    d = small

    def run(with_priors_call):
        ctx = api.Context(0, d["rows"], d["cols"], 2)
        _, converter, aligner = gpu_objects(ctx, "small")
        gref, gcur = api.Cloud(ctx, d["rows"] * d["cols"]), api.Cloud(ctx, d["rows"] * d["cols"])
        converter.compute(gref, d["ref"]); converter.compute(gcur, d["cur"])
        aligner.setReferenceCloud(gref); aligner.setCurrentCloud(gcur)
        aligner.setOuterIterations(1)
        first = aligner.align()
        if with_priors_call:
            aligner.setOuterIterations(10)
            aligner.addRelativePrior(synth.v2t(np.array([0.3, 0.2, -0.2, 0.05, -0.05, 0.05])).astype(np.float32), np.eye(6, dtype=np.float32) * 1e7)
            aligner.align()                                  # projects the reference from ten quite different poses
            aligner.clearPriors()
            aligner.setOuterIterations(1)
        last = aligner.align(images=True)
        imgs = dict(aligner.correspondenceFinder()._images)
        ctx.close()
        return first, last, imgs

    f0, l0, i0 = run(False)
    f1, l1, i1 = run(True)
    assert np.array_equal(f0["T"], f1["T"])
    assert np.array_equal(l0["T"], l1["T"]) and np.array_equal(l0["chi2"], l1["chi2"]) and np.array_equal(l0["K"], l1["K"])
    for k in i0:
        assert np.array_equal(i0[k], i1[k]), k
