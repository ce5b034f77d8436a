"""Size-independent properties of the HIP path at BASELINE's full frame size (VGA batches), no oracle involved:
ground-truth recovery, bitwise invariance to batching / sub-batching / stream count / repetition, forward-backward
consistency of the alignment, projection round trip, converter idempotence."""
import hashlib

import numpy as np
import pytest

from conftest import case_params, make_depth_pair

pytestmark = pytest.mark.gpu

N_PAIRS = 12


@pytest.fixture(scope="module")
def world():
    from g2o_frontend_amd import api, synth
    from test_gpu_parity import gpu_objects
    rows, cols, K, _, _ = case_params("vga")
    ctx = api.Context(0, rows, cols, 16)
    _, converter, aligner = gpu_objects(ctx, "vga")
    pairs = [synth.make_pair(200 + i, rows, cols, K) for i in range(N_PAIRS)]
    refs = [api.Cloud(ctx, rows * cols) for _ in pairs]; curs = [api.Cloud(ctx, rows * cols) for _ in pairs]
    converter.computeBatch(refs + curs, [p[0] for p in pairs] + [p[1] for p in pairs], raw_scale=0.001)
    yield dict(ctx=ctx, converter=converter, aligner=aligner, pairs=pairs, refs=refs, curs=curs, rows=rows, cols=cols, K=K)
    ctx.close()


def _digest(results):
    h = hashlib.sha256()
    for r in results:
        h.update(np.ascontiguousarray(r["T"]).tobytes()); h.update(np.ascontiguousarray(r["chi2"]).tobytes())
        h.update(np.ascontiguousarray(r["C"]).tobytes()); h.update(np.ascontiguousarray(r["K"]).tobytes())
    return h.hexdigest()


def test_every_pair_recovers_the_true_motion(world):
    res = world["aligner"].alignBatch(world["refs"], world["curs"])
    for r, (_, _, Ttrue) in zip(res, world["pairs"]):
        assert np.abs(r["T"][:3, 3] - Ttrue[:3, 3]).max() < 5e-3, np.abs(r["T"][:3, 3] - Ttrue[:3, 3]).max()
        assert np.abs(r["T"][:3, :3] - Ttrue[:3, :3]).max() < 5e-3
        assert np.abs(r["T"][:3, :3] @ r["T"][:3, :3].T - np.eye(3)).max() < 1e-5
        assert r["chi2"][-1] < 0.2 * r["chi2"][0] and r["inliers"] > 100000 and r["iterations"] == 10
        assert np.all(r["C"] <= r["K"]) and np.all(r["K"] <= world["rows"] * world["cols"])


def test_results_do_not_depend_on_batching_streams_or_repetition(world):
    ctx, aligner = world["ctx"], world["aligner"]
    base = _digest(aligner.alignBatch(world["refs"], world["curs"]))
    assert _digest(aligner.alignBatch(world["refs"], world["curs"])) == base              # run to run
    for sub, streams in ((1, 1), (3, 2), (5, 1), (8, 2), (16, 1), (4, 3), (2, 4), (3, 4)):
        ctx.set_subbatch(sub, sub); ctx.set_concurrency(streams)
        assert _digest(aligner.alignBatch(world["refs"], world["curs"])) == base, (sub, streams)
    ctx.set_subbatch(64, 64); ctx.set_concurrency(2)
    single = []
    for a, b in zip(world["refs"], world["curs"]):
        aligner.setReferenceCloud(a); aligner.setCurrentCloud(b)
        single.append(aligner.align())
    assert _digest(single) == base                                                          # one at a time


def test_converter_is_idempotent_and_batch_invariant(world):
    from g2o_frontend_amd import api
    ctx, converter = world["ctx"], world["converter"]
    rows, cols = world["rows"], world["cols"]
    mm = world["pairs"][0][0]
    a, b = api.Cloud(ctx, rows * cols), api.Cloud(ctx, rows * cols)
    converter.computeBatch([a], [mm], raw_scale=0.001)
    depth = ctx.DepthImage_convert_16UC1_to_32FC1(mm)
    converter.compute(b, depth)                                                             # float path, single call
    x, y, z = a.arrays(), b.arrays(), world["refs"][0].arrays()                             # z: converted inside a 24-frame batch
    for k in x:
        assert np.array_equal(x[k].view(np.uint32), y[k].view(np.uint32)) and np.array_equal(x[k].view(np.uint32), z[k].view(np.uint32)), k


def test_frames_from_pinned_host_memory_and_from_hbm_give_the_same_clouds(world):
    """pwn_hip_host_alloc: frames handed over from page-locked host memory (asynchronous copies on the sub-batch streams), from pageable
    memory (the fixture) and from buffers already in HBM are the same input."""
    from g2o_frontend_amd import api
    ctx, converter = world["ctx"], world["converter"]
    rows, cols = world["rows"], world["cols"]
    frames = [p[0] for p in world["pairs"][:5]]
    block = api.pinned_empty((3, rows, cols), np.uint16)                                    # three consecutive frames (one transfer) ...
    pinned = [block[0], block[1], block[2]] + [api.pinned_empty((rows, cols), np.uint16) for _ in frames[3:]]      # ... and two on their own
    for d, f in zip(pinned, frames):
        d[...] = f
    resident = [ctx.upload(f) for f in frames]                                              # pwn_hip_device_alloc + pwn_hip_copy
    assert np.array_equal(resident[3].numpy(), frames[3])
    a = [api.Cloud(ctx, rows * cols) for _ in frames]; b = [api.Cloud(ctx, rows * cols) for _ in frames]
    ctx.set_subbatch(2, 2)                                                                  # three sub-batches over two streams
    converter.computeBatch(a, pinned, raw_scale=0.001)
    converter.computeBatch(b, resident, raw_scale=0.001)
    ctx.set_subbatch(64, 64)
    c = [api.Cloud(ctx, rows * cols) for _ in frames]
    converter.computeBatch(c, pinned, raw_scale=0.001)                                      # one sub-batch: the run of three + two single copies
    for x, y in zip(a, c):
        xa, ya = x.arrays(), y.arrays()
        for k in xa:
            assert np.array_equal(xa[k].view(np.uint32), ya[k].view(np.uint32)), k
    for x, y, z in zip(a, b, world["refs"]):
        xa, ya, za = x.arrays(), y.arrays(), z.arrays()
        for k in xa:
            assert np.array_equal(xa[k].view(np.uint32), ya[k].view(np.uint32)) and np.array_equal(xa[k].view(np.uint32), za[k].view(np.uint32)), k
    # pwn_hip_copy_async: the convert call that follows waits for the queued copies; frames addressed inside one device block
    dev = ctx.upload(np.zeros((3, rows, cols), np.uint16))
    dev.copy_from_async(block)
    e = [api.Cloud(ctx, rows * cols) for _ in range(3)]
    converter.computeBatch(e, [dev.frame(i) for i in range(3)], raw_scale=0.001)
    assert np.array_equal(dev.numpy(), block)
    for x, y in zip(a, e):
        xa, ya = x.arrays(), y.arrays()
        for k in xa:
            assert np.array_equal(xa[k].view(np.uint32), ya[k].view(np.uint32)), k
    dev.free()
    api.pinned_free(block)
    for d in pinned[3:]:
        api.pinned_free(d)
    for d in resident:
        d.free()


def test_forward_backward_consistency(world):
    """align(ref, cur) and align(cur, ref) are inverse motions (up to the accuracy of the registration itself)."""
    aligner = world["aligner"]
    for i in range(3):
        aligner.setReferenceCloud(world["refs"][i]); aligner.setCurrentCloud(world["curs"][i])
        f = aligner.align()["T"].astype(np.float64)
        aligner.setReferenceCloud(world["curs"][i]); aligner.setCurrentCloud(world["refs"][i])
        b = aligner.align()["T"].astype(np.float64)
        assert np.abs(f @ b - np.eye(4)).max() < 3e-3


def test_projection_round_trip_full_size(world):
    """Projecting a cloud from the pose it was unprojected at returns its own index image (every valid pixel its own point)."""
    from g2o_frontend_amd import api
    from test_gpu_parity import gpu_objects
    ctx = world["ctx"]
    proj, converter, _ = gpu_objects(ctx, "vga")
    depth = ctx.DepthImage_convert_16UC1_to_32FC1(world["pairs"][1][0])
    c = api.Cloud(ctx, world["rows"] * world["cols"])
    converter.compute(c, depth)
    idx = converter.indexImage()
    proj.setImageSize(world["rows"], world["cols"]); proj.setTransform(np.eye(4))
    pi, pd = proj.project(c)
    assert np.array_equal(pi, idx)
    valid = idx >= 0
    assert np.array_equal(pd[valid], depth[valid]) and np.all(pd[~valid] == np.finfo(np.float32).max)
    assert c.size() == int(valid.sum()) and np.array_equal(idx[valid], np.arange(valid.sum()))


def test_identical_frames_give_identity(world):
    aligner = world["aligner"]
    aligner.setReferenceCloud(world["refs"][0]); aligner.setCurrentCloud(world["refs"][0])
    r = aligner.align()
    assert np.abs(r["T"] - np.eye(4)).max() < 1e-5 and r["chi2"][-1] < 1e-3 * max(1.0, float(r["inliers"]))


def test_zbuffer_epoch_tags_wrap_without_changing_results():
    """The z-buffer epoch tags run down across calls and the buffers are cleared only when the 12-bit tag space is used up
    (every ~372 alignments of 11 projections): results before, across and after a wrap are bitwise the same, also when the
    image size changes in between (stale words of another geometry must read as empty)."""
    from g2o_frontend_amd import api, synth
    from test_gpu_parity import gpu_objects
    rows, cols, K, _, _ = case_params("small")
    ctx = api.Context(0, 2 * rows, 2 * cols, 4)
    _, converter, aligner = gpu_objects(ctx, "small")
    pairs = [synth.make_pair(300 + i, rows, cols, K) for i in range(3)]
    refs = [api.Cloud(ctx, rows * cols) for _ in pairs]; curs = [api.Cloud(ctx, rows * cols) for _ in pairs]
    converter.computeBatch(refs + curs, [p[0] for p in pairs] + [p[1] for p in pairs], raw_scale=0.001)
    base = _digest(aligner.alignBatch(refs, curs))
    for i in range(800):                      # > 2 wraps of the 4094-tag space at 11 tags per call
        if i % 97 == 5:
            # another image geometry on the same context in between: 2x the size, one pair
            K2 = synth.scaled_K(synth.K_VGA, 2)
            _, conv2, al2 = gpu_objects(ctx, "small")
            conv2.projector().setCameraMatrix([[K2[0], 0, K2[2]], [0, K2[1], K2[3]], [0, 0, 1]]); conv2.projector().setImageSize(2 * rows, 2 * cols)
            al2.projector().setImageSize(2 * rows, 2 * cols); al2.correspondenceFinder().setImageSize(2 * rows, 2 * cols)
            big = synth.make_pair(7, 2 * rows, 2 * cols, K2)
            a, b = api.Cloud(ctx, 4 * rows * cols), api.Cloud(ctx, 4 * rows * cols)
            conv2.computeBatch([a, b], [big[0], big[1]], raw_scale=0.001)
            r = al2.alignBatch([a], [b])[0]
            assert r["iterations"] == 10 and r["inliers"] > 1000
        assert _digest(aligner.alignBatch(refs, curs)) == base, i
    ctx.close()


def test_measured_hbm_bandwidth_is_sane(world):
    """The bandwidth probe bench.py reports next to the spec peak: a streaming read and a copy of 1 GiB land between 2 and 8 TB/s on an
    MI355X (anything else means the probe, not the memory, is broken)."""
    rd, cp = world["ctx"].measure_hbm(1 << 30)
    assert 2000.0 < rd < 8000.0 and 2000.0 < cp < 8000.0, (rd, cp)



def test_pinned_blocks_live_as_long_as_their_views_and_buffers_outlive_their_context():
    """Round-2 advisor findings: (1) a page-locked block (pwn_hip_host_alloc) is released with its LAST numpy view, not by pinned_free while
    views are alive; (2) a DeviceBuffer freed after its context was closed is released (pwn_hip_device_free with a NULL context) instead
    of leaking; (3) data queued with pwn_hip_copy_async is seen by pwn_hip_cloud_upload (every entry point that reads caller pointers absorbs
    the queued copies)."""
    import gc
    from g2o_frontend_amd import api
    a = api.pinned_empty((4, 8), np.float32)
    a[...] = 7.0
    v = a[1]
    api.pinned_free(a)                     # the earlier interface: must not free under the views
    del a; gc.collect()
    assert float(v.sum()) == 56.0          # still mapped
    del v; gc.collect()
    ctx = api.Context(0, 16, 16, 2)
    buf = ctx.upload(np.arange(12, dtype=np.float32))
    ctx.close()
    buf.free()                             # no context any more: plain hipFree inside the library, no error, no leak
    # copy_async -> cloud_upload
    ctx = api.Context(0, 16, 16, 2)
    n = 50
    pts = np.zeros((n, 4), np.float32); pts[:, :3] = np.random.default_rng(0).normal(size=(n, 3)); pts[:, 3] = 1
    host = api.pinned_empty((n, 4), np.float32); host[...] = pts
    dev = ctx.upload(np.zeros((n, 4), np.float32))
    dev.copy_from_async(host)              # queued on the copy stream
    nrm = np.zeros((n, 4), np.float32); curv = np.zeros(n, np.float32); om = np.zeros((n, 16), np.float32)
    c = api.Cloud(ctx, n)
    L = ctx._L
    import ctypes as C
    ctx.check(L.pwn_hip_cloud_upload(ctx.h, c.h, n, C.c_void_p(dev.data_ptr()), nrm.ctypes.data_as(C.c_void_p), curv.ctypes.data_as(C.c_void_p),
                                     om.ctypes.data_as(C.c_void_p), om.ctypes.data_as(C.c_void_p)))
    assert np.array_equal(c.arrays()["points"][:, :3], pts[:, :3])
    dev.free(); ctx.close()


def test_index_shortcut_is_the_projection_it_replaces(world):
    """An alignment takes a cloud's own index image (the converter's) where projecting the cloud would return it: the current cloud always,
    the reference cloud in the first iteration of an identity guess -- single alignments too, whose current z-buffer is then filled in only
    when the finder's images are asked for.  Against the same alignments with the shortcut switched off (pwn_hip_debug_set_index_shortcut:
    every projection executed): poses, traces and counters bitwise equal, single and batch; the finder's four images bitwise equal; the
    matchClouds score equal; and the current index image IS the cloud's own index image."""
    from g2o_frontend_amd import api
    ctx, aligner, refs, curs = world["ctx"], world["aligner"], world["refs"], world["curs"]
    rows, cols = world["rows"], world["cols"]
    keys = ("T", "chi2", "C", "K", "iter_inliers")
    bits = lambda a: np.ascontiguousarray(a).view(np.uint32) if np.asarray(a).dtype.itemsize == 4 else np.ascontiguousarray(a)
    finder = aligner.correspondenceFinder()

    def run(enabled):
        ctx.check(ctx._L.pwn_hip_debug_set_index_shortcut(ctx.h, enabled))
        out = dict(batch=aligner.alignBatch(refs, curs), single=[], images=[], score=[])
        for i in (0, 5, 11):
            aligner.setReferenceCloud(refs[i]); aligner.setCurrentCloud(curs[i])
            out["single"].append(aligner.align())
            m = api.MatchResult()
            ctx.check(ctx._L.pwn_hip_match_score(ctx.h, 50.0, api.C.byref(m)))            # before the images: reads the depths off the cloud when lazy
            out["score"].append((m.image_non_zeros, m.image_outliers, m.image_inliers, np.float32(m.image_reprojection_distance).view(np.uint32)))
            aligner.align(images=True)
            out["images"].append({k: v.copy() for k, v in finder._images.items()})
            m2 = api.MatchResult()
            ctx.check(ctx._L.pwn_hip_match_score(ctx.h, 50.0, api.C.byref(m2)))           # after them: from the z-buffer the images call filled in
            assert (m2.image_non_zeros, m2.image_outliers, m2.image_inliers) == out["score"][-1][:3]
        return out

    try:
        off = run(0)
        on = run(1)
    finally:
        ctx.check(ctx._L.pwn_hip_debug_set_index_shortcut(ctx.h, 1))
    for a, b in zip(on["batch"], off["batch"]):
        for k in keys:
            assert np.array_equal(bits(a[k]), bits(b[k])), k
    for j, (a, b) in enumerate(zip(on["single"], off["single"])):
        for k in keys:
            assert np.array_equal(bits(a[k]), bits(b[k])), (j, k)
        for k in keys:                                                      # and the single alignment is the batch entry
            assert np.array_equal(bits(a[k]), bits(on["batch"][(0, 5, 11)[j]][k])), (j, k)
    for j, (a, b) in enumerate(zip(on["images"], off["images"])):
        for k in ("ref_index", "ref_depth", "cur_index", "cur_depth"):
            assert np.array_equal(bits(a[k]), bits(b[k])), (j, k)
        assert (a["cur_index"] >= 0).sum() == curs[(0, 5, 11)[j]].size()
    assert on["score"] == off["score"]
    # the current index image of the finder is the converter's index image of that cloud
    conv = world["converter"]
    c = api.Cloud(ctx, rows * cols)
    conv.compute(c, np.asarray(world["pairs"][5][1], np.float32) * np.float32(0.001))
    assert np.array_equal(conv.indexImage(), on["images"][1]["cur_index"])


def test_single_point_projector_forms_are_the_kernels_expressions():
    """PinholePointProjector::project(x, y, f, p) / unProject(p, x, y, d) / projectInterval (host code of the library) against what the kernels wrote for
    the same pixels and points: the converter's points and interval image, and the index / depth images of a projection under a non-identity camera pose."""
    from g2o_frontend_amd import api
    from conftest import case_params, make_depth_pair
    from test_gpu_parity import gpu_objects
    name = "small"
    rows, cols, K, conv, alig = case_params(name)
    ref, _, _, _, _ = make_depth_pair(name, 6)
    ctx = api.Context(0, rows, cols, 2)
    proj, converter, _ = gpu_objects(ctx, name)
    cloud = api.Cloud(ctx, rows * cols)
    converter.compute(cloud, ref)
    idx, itv = converter.indexImage(), converter.intervalImage()
    P = cloud.arrays()["points"][:, :3]
    rng = np.random.default_rng(3)
    pix = rng.integers(0, rows * cols, 400)
    seen = 0
    for q in pix:
        r, c = divmod(int(q), cols)
        ok, p = proj.unProjectPixel(c, r, ref[r, c])
        assert ok == (idx[r, c] >= 0)
        assert proj.projectInterval(c, r, ref[r, c], conv["world_radius"]) == itv[r, c]
        if ok:
            assert np.array_equal(p.view(np.uint32), np.ascontiguousarray(P[idx[r, c]]).view(np.uint32)), (r, c)
            seen += 1
    assert seen > 300
    # the per-field accessors of pwn::Cloud (cloud.h:33-131) are the fields of arrays()
    A = cloud.arrays()
    for name, got in (("points", cloud.points()), ("normals", cloud.normals()), ("curvature", cloud.curvatures()), ("omega_p", cloud.pointInformationMatrix()),
                      ("omega_n", cloud.normalInformationMatrix())):
        assert np.array_equal(got.view(np.uint32), A[name].view(np.uint32)), name
    assert cloud.traversabilityVector() == []
    # a projection from another pose: the pixel and depth a point gets here are where the kernel put it -- unless a nearer point won that pixel
    from g2o_frontend_amd import synth
    proj.setTransform(synth.v2t(np.array([0.03, -0.02, 0.04, 0.01, -0.012, 0.015])).astype(np.float32))
    pidx, pdep = proj.project(cloud)
    won = 0
    for i in rng.integers(0, len(P), 400):
        ok, x, y, d = proj.projectPoint(P[i])
        if not ok or not (0 <= x < cols and 0 <= y < rows):
            continue
        assert pidx[y, x] >= 0 and pdep[y, x] <= np.float32(d)
        if pidx[y, x] == i:
            assert np.float32(d).view(np.uint32) == pdep[y, x].view(np.uint32)
            won += 1
    assert won > 250
    ctx.close()


def test_linearizer_T_after_align_is_the_inverse_of_the_result():
    """Aligner::align leaves the linearizer with the inverse of the final transform (aligner.cpp:90 per iteration, :165-167 in _computeStatistics):
    code ported from the reference that reads linearizer()->T() after align gets that, not the identity it was constructed with."""
    from g2o_frontend_amd import api
    from test_gpu_parity import gpu_objects
    rows, cols, K, conv, alig = case_params("small")
    ref, cur, _, _, _ = make_depth_pair("small", 2)
    ctx = api.Context(0, rows, cols, 2)
    try:
        _, converter, aligner = gpu_objects(ctx, "small")
        a, b = api.Cloud(ctx, rows * cols), api.Cloud(ctx, rows * cols)
        converter.compute(a, ref); converter.compute(b, cur)
        assert np.array_equal(aligner.linearizer().T(), np.eye(4, dtype=np.float32))
        aligner.setReferenceCloud(a); aligner.setCurrentCloud(b)
        g = aligner.align()
        want = api.iso_inverse(g["T"]); want[3] = (0, 0, 0, 1)
        assert np.array_equal(aligner.linearizer().T().view(np.uint32), want.view(np.uint32))
        assert not np.array_equal(want, np.eye(4, dtype=np.float32))
    finally:
        ctx.close()
