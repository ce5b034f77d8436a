"""One candidate batch as ONE submission (pwn_hip_convert_align_batch_u16) and the result records packed on the device
(pwn_hip_align_batch_records; include/pwn_hip.h: PWN_HIP_RECORD_FLOATS).  The reference converts and aligns the candidates of a closure one
after the other (pwn_tracker/pwn_closer.cpp:92-111, pwn_matcher_base.cpp:77-85,120-128); here the conversion of a sub-batch is queued in
front of its alignment on the sub-batch's stream.  What must hold:
  * results bit for bit those of convert_batch_u16 followed by align_batch (any sub-batch size, one or two streams, host or device frames),
    and therefore the oracle's (teacher-forced chi2 1e-5, exact counters: checked on a sampled pair);
  * the records the kernel writes == shard.pack_results_raw of the host results, bit for bit, in a device buffer and in host memory;
  * clouds keep their sizes / index images: a later single alignment of a pair equals its batch result.
"""
import numpy as np
import pytest

from conftest import case_params

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def _rig(name, n, max_batch, omega="exact9"):
    from g2o_frontend_amd import api, synth
    from test_gpu_parity import gpu_objects
    rows, cols, K, _, _ = case_params(name)
    ctx = api.Context(0, rows, cols, max_batch, omega_storage=omega)
    _, converter, aligner = gpu_objects(ctx, name)
    pairs = [synth.make_pair(7000 + s, rows, cols, K) for s in range(n)]
    refs = [api.Cloud(ctx, rows * cols) for _ in range(n)]; curs = [api.Cloud(ctx, rows * cols) for _ in range(n)]
    return ctx, converter, aligner, pairs, refs, curs


@pytest.mark.parametrize("sub_frames,sub_pairs,streams,device_frames", [(64, 64, 2, True), (4, 3, 2, True), (5, 4, 1, False), (16, 8, 2, False), (3, 5, 2, True)])
def test_fused_step_equals_convert_then_align(sub_frames, sub_pairs, streams, device_frames):
    from g2o_frontend_amd import shard
    n = 11
    ctx, converter, aligner, pairs, refs, curs = _rig("small", n, 32)
    rf = [p[0] for p in pairs]; cf = [p[1] for p in pairs]
    # reference sequence: two calls with a host wait between them
    ctx.set_subbatch(64, 64); ctx.set_concurrency(2)
    converter.computeBatch(refs + curs, rf + cf, raw_scale=0.001)
    want = aligner.alignBatch(refs, curs, raw=True).copy()
    sizes = [(r.size(), c.size()) for r, c in zip(refs, curs)]
    # one submission, other clouds
    from g2o_frontend_amd import api
    rows, cols, _, _, _ = case_params("small")
    refs2 = [api.Cloud(ctx, rows * cols) for _ in range(n)]; curs2 = [api.Cloud(ctx, rows * cols) for _ in range(n)]
    ctx.set_subbatch(sub_frames, sub_pairs); ctx.set_concurrency(streams)
    bufs = [ctx.upload(f) for f in rf + cf] if device_frames else None
    frames = bufs if device_frames else rf + cf
    rec = np.full((n, shard.RECORD_FLOATS), -7.0, np.float32)
    ids = np.arange(100, 100 + n, dtype=np.int32)
    got = aligner.convertAlignBatch(converter, refs2, curs2, frames[:n], frames[n:], raw_scale=0.001, records=rec, pair_ids=ids)
    for k in ("T", "chi2", "iter_inliers", "iter_correspondences", "iter_candidates", "error", "inliers", "iterations", "n_reference", "n_current"):
        assert np.array_equal(_bits(got[k]) if got[k].dtype == np.float32 else got[k], _bits(want[k]) if want[k].dtype == np.float32 else want[k]), k
    assert [(r.size(), c.size()) for r, c in zip(refs2, curs2)] == sizes
    assert np.array_equal(_bits(rec), _bits(shard.pack_results_raw(got, ids)))
    # the clouds are complete: a later single alignment of a pair is its batch result
    aligner.setReferenceCloud(refs2[7]); aligner.setCurrentCloud(curs2[7])
    g = aligner.align()
    assert np.array_equal(_bits(g["chi2"]), _bits(got["chi2"][7][:10])) and np.array_equal(_bits(np.asarray(g["T"], np.float32).T.reshape(-1)), _bits(got["T"][7]))
    a, b = refs[3].arrays(), refs2[3].arrays()
    for k in a:
        assert np.array_equal(_bits(a[k]), _bits(b[k])), k
    if bufs:
        for f in bufs:
            f.free()
    ctx.close()


def test_records_on_the_device_and_without_results():
    from g2o_frontend_amd import shard
    n = 6
    ctx, converter, aligner, pairs, refs, curs = _rig("small", n, 16)
    converter.computeBatch(refs + curs, [p[0] for p in pairs] + [p[1] for p in pairs], raw_scale=0.001)
    want = aligner.alignBatch(refs, curs, raw=True).copy()
    rec = ctx.upload(np.full((n, shard.RECORD_FLOATS), -3.0, np.float32))            # a device buffer, as the tensor an all-gather sends
    got = aligner.alignBatchRecords(refs, curs, rec, first_pair_id=40)
    assert np.array_equal(_bits(got["chi2"]), _bits(want["chi2"]))
    assert np.array_equal(_bits(rec.numpy()), _bits(shard.pack_results_raw(want, np.arange(40, 40 + n))))
    assert aligner.alignBatchRecords(refs, curs, rec, pair_ids=np.arange(n)[::-1].copy(), want_results=False) is None      # records only
    assert np.array_equal(_bits(rec.numpy()), _bits(shard.pack_results_raw(want, np.arange(n)[::-1])))
    # fewer iterations than trace slots, and none: the words past the last iteration stay 0
    aligner.setOuterIterations(3)
    h = np.zeros((n, shard.RECORD_FLOATS), np.float32)
    r3 = aligner.alignBatchRecords(refs, curs, h)
    assert np.array_equal(_bits(h), _bits(shard.pack_results_raw(r3, np.arange(n)))) and (h[:, 23:30] == 0).all() and (h[:, 62] == 3).all()
    aligner.setOuterIterations(0)
    r0 = aligner.alignBatchRecords(refs, curs, h)
    assert np.array_equal(_bits(h), _bits(shard.pack_results_raw(r0, np.arange(n)))) and (h[:, 16:19] == 0).all()
    rec.free()
    ctx.close()


def test_fused_step_vga_against_oracle_sym6(oracle):
    """40 VGA pairs in the bench configuration (sub-batches of 16 on two streams, sym6 clouds, device frames): equal to the two-call
    sequence; a sampled pair teacher-forced against the oracle"""
    from g2o_frontend_amd import api
    from test_gpu_parity import oracle_params, _check_teacher_forced
    from test_omega_sym6 import compare_clouds_sym6
    n = 40
    ctx, converter, aligner, pairs, refs, curs = _rig("vga", n, 64, omega="sym6")
    rows, cols, _, _, _ = case_params("vga")
    ctx.set_subbatch(32, 16); ctx.set_concurrency(2)
    bufs = [ctx.upload(p[0]) for p in pairs] + [ctx.upload(p[1]) for p in pairs]
    got = aligner.convertAlignBatch(converter, refs, curs, bufs[:n], bufs[n:], raw_scale=0.001).copy()
    again = aligner.convertAlignBatch(converter, refs, curs, bufs[:n], bufs[n:], raw_scale=0.001)
    assert np.array_equal(_bits(got["chi2"]), _bits(again["chi2"])) and np.array_equal(_bits(got["T"]), _bits(again["T"]))
    refs2 = [api.Cloud(ctx, rows * cols) for _ in range(n)]; curs2 = [api.Cloud(ctx, rows * cols) for _ in range(n)]
    converter.computeBatch(refs2 + curs2, bufs, raw_scale=0.001)
    want = aligner.alignBatch(refs2, curs2, raw=True)
    for k in ("T", "chi2"):
        assert np.array_equal(_bits(got[k]), _bits(want[k])), k
    for k in ("iter_inliers", "iter_correspondences", "iter_candidates", "n_reference", "n_current"):
        assert np.array_equal(got[k], want[k]), k
    i = 23
    cp, ap = oracle_params(oracle, "vga", accumulate_fp64=1)
    oref, _, _ = oracle.convert(cp, oracle.convert_16u_to_32f(pairs[i][0])); ocur, _, _ = oracle.convert(cp, oracle.convert_16u_to_32f(pairs[i][1]))
    compare_clouds_sym6(oref.arrays(), refs[i].arrays()); compare_clouds_sym6(ocur.arrays(), curs[i].arrays())
    o = oracle.align(ap, oref, ocur)
    aligner.setReferenceCloud(refs[i]); aligner.setCurrentCloud(curs[i])
    worst = _check_teacher_forced(aligner, o)
    print(f"fused step, pair {i}: worst teacher-forced chi2 rel diff {worst:.1e}")
    for f in bufs:
        f.free()
    ctx.close()


def test_the_shipped_configuration_against_the_oracle(oracle):
    """The exact configuration bench.py times and reports (BENCH_rNN.json): 128 VGA pairs, omega_storage = sym6, ONE submission per step
    (pwn_hip_convert_align_batch_u16), the library's default plan (four streams, 4 x 32 pairs), uint16 frames resident on the device, result
    records packed on the device into a CUDA tensor.  Held against
      * the oracle on sampled pairs of every sub-batch: converter clouds (sym6 comparison: everything bit for bit but the mirrored lower
        triangle of Omega_p, 1e-6), every iteration of the oracle's trace re-run from the oracle's iterate (teacher-forced chi2 <= 1e-5 vs the
        fp64-accumulated oracle, K_i / C_i / inliers_i exact), final pose within 1e-5 of the oracle's own free run;
      * the two-call path (convert_batch_u16, then align_batch_records) on other clouds: the same records, bit for bit."""
    from g2o_frontend_amd import api, shard
    from test_gpu_parity import oracle_params, _check_teacher_forced
    from test_omega_sym6 import compare_clouds_sym6
    n = 128
    ctx, converter, aligner, pairs, refs, curs = _rig("vga", n, 256, omega="sym6")      # 4 streams x 64 slots, as bench.py's context
    rows, cols, _, _, _ = case_params("vga")
    ctx.set_subbatch(64, 64); ctx.set_concurrency(4)                                     # bench.py's defaults (--sub-frames 64 --sub-pairs 64 --streams 4)
    bufs = [ctx.upload(p[0]) for p in pairs] + [ctx.upload(p[1]) for p in pairs]
    rec = ctx.upload(np.full((n, shard.RECORD_FLOATS), -7.0, np.float32))                # a device buffer, as the tensor the all-gather sends
    ids = np.arange(9000, 9000 + n, dtype=np.int32)
    prep = aligner.convertAlignHandles(refs, curs, bufs[:n], bufs[n:], converter=converter)
    ctx.set_profiling(True)
    got = aligner.convertAlignBatch(converter, None, None, None, None, raw_scale=0.001, records=rec, pair_ids=ids, prepared=prep).copy()
    launches = ctx.stage_ms("corr_linearize")[1]
    ctx.set_profiling(False)
    assert launches == 4 * 10, launches                                                  # the plan the line reports: 4 sub-batches of 32 pairs x 10 iterations
    rec_fused = rec.numpy().copy()
    assert np.array_equal(_bits(rec_fused), _bits(shard.pack_results_raw(got, ids)))
    # the two-call path on other clouds
    refs2 = [api.Cloud(ctx, rows * cols) for _ in range(n)]; curs2 = [api.Cloud(ctx, rows * cols) for _ in range(n)]
    converter.computeBatch(refs2 + curs2, bufs, raw_scale=0.001)
    rec2 = ctx.upload(np.full((n, shard.RECORD_FLOATS), -3.0, np.float32))
    aligner.alignBatchRecords(refs2, curs2, rec2, pair_ids=ids, want_results=False)
    assert np.array_equal(_bits(rec2.numpy()), _bits(rec_fused))
    # the oracle on one pair of every sub-batch (first, inner, last positions)
    cp, ap = oracle_params(oracle, "vga", accumulate_fp64=1)
    worst_chi2 = worst_pose = worst_lower = 0.0
    for i in (0, 45, 77, 127):
        oref, _, _ = oracle.convert(cp, oracle.convert_16u_to_32f(pairs[i][0])); ocur, _, _ = oracle.convert(cp, oracle.convert_16u_to_32f(pairs[i][1]))
        worst_lower = max(worst_lower, compare_clouds_sym6(oref.arrays(), refs[i].arrays()), compare_clouds_sym6(ocur.arrays(), curs[i].arrays()))
        o = oracle.align(ap, oref, ocur)
        aligner.setReferenceCloud(refs[i]); aligner.setCurrentCloud(curs[i])
        worst_chi2 = max(worst_chi2, _check_teacher_forced(aligner, o))                  # asserts 1e-5 and exact counters per iteration
        it0 = o["iterations"][0]
        assert (int(got["iter_candidates"][i][0]), int(got["iter_correspondences"][i][0]), int(got["iter_inliers"][i][0])) == (it0["K"], it0["C"], it0["inliers"]), i
        T = got["T"][i].reshape(4, 4).T
        d = float(np.abs(T - o["T"]).max()); worst_pose = max(worst_pose, d)
        assert d <= 1e-5, (i, d)
        assert int(got["n_reference"][i]) == len(oref) and int(got["n_current"][i]) == len(ocur)
    print(f"shipped configuration (128 VGA pairs, sym6, one submission, 4 x 32): 4 pairs vs oracle: worst teacher-forced chi2 rel diff {worst_chi2:.1e}, "
          f"worst |T - T_oracle| {worst_pose:.1e}, mirrored lower triangle within {worst_lower:.1e} of |Omega_p|; records == two-call path bitwise")
    ctx.close()


def test_fused_step_error_paths():
    from g2o_frontend_amd._lib import PwnHipError
    ctx, converter, aligner, pairs, refs, curs = _rig("small", 2, 8)
    big = np.zeros((240, 320), np.uint16)
    with pytest.raises(PwnHipError):                      # frames larger than the context
        aligner.convertAlignBatch(converter, refs, curs, [big, big], [big, big])
    with pytest.raises(PwnHipError):                      # neither results nor records
        aligner.convertAlignBatch(converter, refs, curs, [p[0] for p in pairs], [p[1] for p in pairs], want_results=False)
    import ctypes as C
    empty = ((C.c_void_p * 0)(), (C.c_void_p * 0)(), (C.c_void_p * 0)(), (C.c_void_p * 0)(), 0, (120, 160))
    assert len(aligner.convertAlignBatch(converter, None, None, None, None, prepared=empty)) == 0             # an empty batch is a no-op
    with pytest.raises(PwnHipError) as e:                 # a cloud that appears twice among the 2 n clouds of a step: two streams would write it at once
        aligner.convertAlignBatch(converter, [refs[0], refs[0]], curs, [p[0] for p in pairs], [p[1] for p in pairs])
    assert e.value.code == 1
    with pytest.raises(PwnHipError):
        aligner.convertAlignBatch(converter, refs, [refs[1], curs[1]], [p[0] for p in pairs], [p[1] for p in pairs])
    with pytest.raises(ValueError):                       # a records buffer smaller than n x 64 floats never reaches the library
        aligner.convertAlignBatch(converter, refs, curs, [p[0] for p in pairs], [p[1] for p in pairs], records=np.empty((1, 64), np.float32))
    ctx.wait_stream(0)                                    # ordering against a caller's stream (here: the legacy default stream, idle): a no-op that must not fail
    r = aligner.convertAlignBatch(converter, refs, curs, [p[0] for p in pairs], [p[1] for p in pairs])      # the context still works
    assert (r["iterations"] == 10).all()
    ctx.close()


def test_step_refuses_frames_beyond_the_aligners_index_field():
    """The aligner's 32-bit z-buffer word indexes 2^21 points.  In the one-submission step the clouds' host-side sizes are those of their previous
    content while the call is queued, so the step bounds what the conversion can produce: frames of more than 2^21 pixels are refused with
    PWN_HIP_ERR_CAPACITY (the two-call path refuses in align_batch, which sees the converted sizes)."""
    from g2o_frontend_amd import api
    from g2o_frontend_amd._lib import PwnHipError
    from test_gpu_parity import gpu_objects
    rows, cols = 1456, 1456                                # 2 119 936 pixels > 2^21
    ctx = api.Context(0, rows, cols, 2)
    _, converter, aligner = gpu_objects(ctx, "vga")
    converter.projector().setImageSize(rows, cols)
    aligner.projector().setImageSize(rows, cols); aligner.correspondenceFinder().setImageSize(rows, cols)
    a, b = api.Cloud(ctx, rows * cols), api.Cloud(ctx, rows * cols)
    f = np.full((rows, cols), 1500, np.uint16)
    with pytest.raises(PwnHipError) as e:
        aligner.convertAlignBatch(converter, [a], [b], [f], [f])
    assert e.value.code == 6
    ctx.close()
