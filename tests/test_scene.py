"""Scene maintenance: sensor-noise Gaussians of unProject, Cloud::add, Merger::merge,
VoxelCalculator::compute and Cloud::save/load.

CPU: the oracle's restatement against independent numpy models and its own invariants.
GPU: the HIP path through the C-ABI against the oracle -- every array bit-exact (the merge accumulates each target's
information matrices in ascending point index, the order of the reference's sequential loop), files byte-identical.
"""
import os

import numpy as np
import pytest

from conftest import case_params, make_depth_pair


def _bits(a):
    """bit pattern of a float array with -0.0 folded onto +0.0 (the reference's 4x4 products add exact-zero fourth terms, which can
    turn a -0.0 into +0.0; the 3x3 products of the kernels do not: the sign of a zero is the only thing that may differ)"""
    a = np.ascontiguousarray(a)
    if a.dtype != np.float32:
        return a
    a = a.copy(); a[a == 0] = 0
    return a.view(np.uint32)


def _K3(K):
    return np.array([[K[0], 0, K[2]], [0, K[1], K[3]], [0, 0, 1]], np.float64)


OFFSET = np.array([[0, 0, 1, 0.1], [-1, 0, 0, 0.02], [0, -1, 0, 0.3], [0, 0, 0, 1]], np.float32)   # a typical sensor mounting


def _oracle_scene(oracle, name="small", seed=1, sensor_offset=None, n_views=2):
    """scene = view 0 + view 1 moved by the true motion (Cloud::add), Gaussians on"""
    rows, cols, K, conv, _ = case_params(name)
    ref, cur, Ttrue, _, _ = make_depth_pair(name, seed)
    oracle.set_gaussians(True)
    try:
        cp = oracle.converter_params(K=K, sensor_offset=sensor_offset, **conv)
        clouds = [oracle.convert(cp, d)[0] for d in (ref, cur)[:n_views]]
    finally:
        oracle.set_gaussians(False)
    return rows, cols, K, conv, (ref, cur), Ttrue, clouds


# ------------------------------------------------------------------------------------------------- CPU: oracle
def test_oracle_gaussians_match_a_float64_model(oracle):
    rows, cols, K, conv, (ref, _), _, (c0, _) = _oracle_scene(oracle)
    g = c0.gaussians(); a = c0.arrays()
    assert len(g["mean"]) == len(a["points"]) and np.all(g["flags"] == 1)
    assert np.array_equal(g["mean"], a["points"][:, :3])                       # Gaussian3f(point.head<3>(), cov)
    iK = np.linalg.inv(_K3(K))
    rr, cc = np.nonzero((ref >= conv["min_distance"]) & (ref <= conv["max_distance"]))
    z = ref[rr, cc].astype(np.float64)
    fB = 0.075 * K[0]; alpha = 0.1
    for i in np.linspace(0, len(z) - 1, 50).astype(int):
        J = iK @ np.array([[z[i], 0, cc[i]], [0, z[i], rr[i]], [0, 0, 1.0]])
        want = J @ np.diag([3.0, 3.0, alpha * z[i] ** 2 / (fB + z[i] * alpha)]) @ J.T
        got = g["cov"][i].reshape(3, 3).T
        assert np.allclose(got, want, rtol=2e-5, atol=1e-9)
    # with a sensor offset: R cov R^T, R mean + t (gaussian3.h:65-73)
    _, _, _, _, _, _, (c1, _) = _oracle_scene(oracle, sensor_offset=OFFSET)
    g1 = c1.gaussians()
    R = OFFSET[:3, :3].astype(np.float64); t = OFFSET[:3, 3].astype(np.float64)
    assert np.allclose(g1["mean"], g["mean"].astype(np.float64) @ R.T + t, atol=1e-5)
    C0 = g["cov"].reshape(-1, 3, 3).transpose(0, 2, 1).astype(np.float64)
    C1 = g1["cov"].reshape(-1, 3, 3).transpose(0, 2, 1)
    assert np.allclose(C1, R @ C0 @ R.T, rtol=1e-4, atol=1e-8)
    assert np.allclose(g1["mean"], c1.arrays()["points"][:, :3], atol=1e-6)


def test_oracle_merge_invariants(oracle):
    rows, cols, K, conv, _, Ttrue, (c0, c1) = _oracle_scene(oracle)
    scene = oracle.Cloud()
    scene.add(c0, np.eye(4)); scene.add(c1, Ttrue)
    n0 = len(scene)
    assert n0 == len(c0) + len(c1) and scene.num_gaussians() == n0
    before = scene.arrays(); gb = scene.gaussians()
    k, col = oracle.merge(scene, K, np.eye(4), conv["min_distance"], conv["max_distance"], rows, cols)
    idx = np.arange(n0)
    winners = col == idx; merged = (col >= 0) & ~winners; untouched = col < 0
    assert k == int(winners.sum() + untouched.sum()) == len(scene) and merged.sum() > 1000
    assert np.all(winners[col[merged]])                                         # a target is the z-buffer winner of its pixel
    assert scene.num_gaussians() == n0                                          # merger.cpp:108-112: the Gaussian vector is not resized
    after = scene.arrays(); ga = scene.gaussians()
    keep = np.nonzero(winners | untouched)[0]
    assert np.array_equal(after["normals"], before["normals"][keep]) and np.array_equal(after["omega_p"], before["omega_p"][keep])
    # untouched points keep their coordinates; a winner moves to the information-weighted mean of what merged into it
    pos = {int(i): j for j, i in enumerate(keep)}
    u = np.nonzero(untouched)[0][:200]
    assert np.array_equal(after["points"][[pos[int(i)] for i in u]], before["points"][u])
    C = gb["cov"].reshape(-1, 3, 3).transpose(0, 2, 1).astype(np.float64)
    targets, counts = np.unique(col[merged], return_counts=True)
    for t in targets[np.argsort(-counts)][:20]:
        members = np.concatenate([[t], np.nonzero(merged & (col == t))[0]])
        infos = np.linalg.inv(C[members])
        want = np.linalg.solve(infos.sum(0), np.einsum("nij,nj->i", infos, gb["mean"][members].astype(np.float64)))
        assert np.allclose(after["points"][pos[int(t)], :3], want, atol=2e-3)
        assert ga["flags"][pos[int(t)]] == 3
    # the fused points do not run away: still near the members' positions
    assert np.abs(after["points"][:, :3]).max() < 10


def test_oracle_voxelize_canonical_vs_literal(oracle):
    _, _, _, _, _, Ttrue, (c0, c1) = _oracle_scene(oracle)
    def scene():
        s = oracle.Cloud(); s.add(c0, np.eye(4)); s.add(c1, Ttrue); return s
    res = 0.05
    s = scene(); pts = s.arrays()["points"]
    k, kept = oracle.voxelize(s, res, literal=False)
    keys = (pts[:, :3] * np.float32(1.0 / res)).astype(np.int32)                # truncation like (int)(p * inverseResolution)
    _, first = np.unique(keys, axis=0, return_index=True)                       # np.unique sorts rows lexicographically
    assert k == len(first) and np.array_equal(kept, first)
    assert np.array_equal(s.arrays()["points"], pts[first])
    s2 = scene()
    k2, kept2 = oracle.voxelize(s2, res, literal=True)
    # the reference's comparator is not a strict weak ordering: the map can miss existing voxels, never merge distinct ones
    assert set(first.tolist()) <= set(kept2.tolist()) and k2 >= k
    assert len(np.unique(keys[kept2], axis=0)) == k


def test_oracle_save_load_round_trip(oracle, tmp_path):
    _, _, _, _, _, Ttrue, (c0, _) = _oracle_scene(oracle)
    a = c0.arrays(stats=True)
    fb, ft = tmp_path / "c.pwn", tmp_path / "c.txt"
    assert c0.save(fb, Ttrue, 1, True) and c0.save(ft, Ttrue, 7, False)
    b, Tb = oracle.Cloud.load(fb)
    assert b is not None and len(b) == len(c0)
    bb = b.arrays(stats=True)
    for k in ("points", "normals", "stats", "eigenvalues", "npoints", "curvature"):
        assert np.array_equal(_bits(bb[k]), _bits(a[k])), k
    assert np.allclose(Tb, Ttrue, atol=1e-5)                                    # stored as t2v(T) with 6 significant digits
    head = open(ft).read().split("\n")
    assert head[0] == f"PWNCLOUD {len(c0) // 7} 0" and head[2].startswith("POINTWITHSTATS ") and len(head[2].split()) == 23
    t, Tt = oracle.Cloud.load(ft)
    assert t is not None and len(t) == len(c0) // 7
    assert np.allclose(t.arrays()["points"], a["points"][::7][:len(t)], rtol=1e-5, atol=1e-6)
    assert os.path.getsize(fb) == len(open(fb, "rb").read().split(b"\n", 2)[0]) + len(open(fb, "rb").read().split(b"\n", 2)[1]) + 2 + 176 * len(c0)


# ------------------------------------------------------------------------------------------------- GPU parity
def _gpu_convert(ctx, name, depths, sensor_offset=None):
    from g2o_frontend_amd import api
    from test_gpu_parity import gpu_objects
    rows, cols, _, _, _ = case_params(name)
    proj, converter, _ = gpu_objects(ctx, name)
    out = []
    for d in depths:
        c = api.Cloud(ctx, rows * cols)
        converter.compute(c, d, sensorOffset=sensor_offset, keep_stats=True, gaussians=True)
        out.append(c)
    return proj, converter, out


def _same_cloud(o, g, gauss=True, stats=True):
    oa, ga = o.arrays(stats=stats), g.arrays(stats=stats)
    assert len(o) == g.size()
    for k in oa:
        assert np.array_equal(_bits(oa[k]), _bits(ga[k])), k
    if gauss:
        og, gg = o.gaussians(), g.gaussians()
        assert o.num_gaussians() == g.numGaussians()
        assert np.array_equal(og["flags"], gg["flags"])
        m, i = (og["flags"] & 1) != 0, (og["flags"] & 2) != 0
        assert np.array_equal(_bits(og["mean"][m]), _bits(gg["mean"][m])) and np.array_equal(_bits(og["cov"][m]), _bits(gg["cov"][m]))
        assert np.array_equal(_bits(og["info"][i]), _bits(gg["info"][i])) and np.array_equal(_bits(og["info_vec"][i]), _bits(gg["info_vec"][i]))


@pytest.fixture(scope="module")
def gctx():
    from g2o_frontend_amd import api
    c = api.Context(device=0, max_rows=480, max_cols=640, max_batch=2, omega_storage="exact9")
    yield c
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("offset", [None, OFFSET])
def test_gaussians_bit_exact(gctx, oracle, offset):
    rows, cols, K, conv, (ref, cur), _, oclouds = _oracle_scene(oracle, sensor_offset=offset)
    _, _, gclouds = _gpu_convert(gctx, "small", (ref, cur), offset)
    for o, g in zip(oclouds, gclouds):
        _same_cloud(o, g)


@pytest.mark.gpu
@pytest.mark.parametrize("name,offset", [("small", None), ("small", OFFSET), ("vga", None)])
def test_add_merge_voxelize_bit_exact(gctx, oracle, name, offset):
    """pwn_aligner.cpp:205-208: scene->add(cloud, T); merger.merge(scene, T * sensorOffset) -- two rounds, then a voxel grid"""
    from g2o_frontend_amd import api
    rows, cols, K, conv, depths, Ttrue, (o0, o1) = _oracle_scene(oracle, name, 1 if name == "small" else 0, sensor_offset=offset)
    proj, converter, (g0, g1) = _gpu_convert(gctx, name, depths, offset)
    so = np.eye(4, dtype=np.float32) if offset is None else offset
    oscene = oracle.Cloud(); gscene = api.Cloud(gctx, 3 * rows * cols)
    merger = api.Merger(); merger.setDepthImageConverter(converter); merger.setImageSize(rows, cols)
    view = np.eye(4, dtype=np.float32)
    for rnd, (oc, gc, T) in enumerate(((o0, g0, np.eye(4, dtype=np.float32)), (o1, g1, Ttrue), (o0, g0, Ttrue @ Ttrue))):
        oscene.add(oc, T); gscene.add(gc, T)
        _same_cloud(oscene, gscene)
        pose = (view if rnd == 0 else T) @ so
        ok, ocol = oracle.merge(oscene, K, pose, conv["min_distance"], conv["max_distance"], rows, cols)
        gk = merger.merge(gscene, pose)
        assert gk == ok and np.array_equal(merger.collapsedIndices(), ocol), rnd
        _same_cloud(oscene, gscene)
        assert rnd == 0 or ((ocol >= 0) & (ocol != np.arange(len(ocol)))).sum() > 1000      # something did merge
    vox = api.VoxelCalculator()
    ok, okept = oracle.voxelize(oscene, 0.03, literal=False)
    gk = vox.compute(gscene, 0.03)
    assert gk == ok and np.array_equal(vox.keptIndices(), okept)
    # voxelcalculator.cpp:62-64: the Gaussians survive only when their vector has the cloud's size (not after a merge)
    assert oscene.num_gaussians() == gscene.numGaussians() == 0
    _same_cloud(oscene, gscene, gauss=False)


@pytest.mark.gpu
def test_cloud_files_byte_identical(gctx, oracle, tmp_path):
    from g2o_frontend_amd import api
    rows, cols, K, conv, depths, Ttrue, (o0, _) = _oracle_scene(oracle)
    _, _, (g0,) = _gpu_convert(gctx, "small", depths[:1])
    for binary, step in ((True, 1), (False, 1), (False, 5), (True, 3)):
        fo, fg = tmp_path / f"o_{binary}_{step}.pwn", tmp_path / f"g_{binary}_{step}.pwn"
        assert o0.save(fo, Ttrue, step, binary) and g0.save(fg, Ttrue, step, binary)
        assert open(fo, "rb").read() == open(fg, "rb").read(), (binary, step)
    # load what the other side wrote
    c = api.Cloud(gctx, rows * cols)
    T = c.load(tmp_path / "o_True_1.pwn")
    ob, To = oracle.Cloud.load(tmp_path / "g_True_1.pwn")
    assert np.array_equal(T, To)
    a, b = c.arrays(stats=True), ob.arrays(stats=True)
    for k in ("points", "normals", "curvature", "stats", "eigenvalues", "npoints"):
        assert np.array_equal(_bits(a[k]), _bits(b[k])), k
    c2 = api.Cloud(gctx, rows * cols)
    c2.load(tmp_path / "o_False_5.pwn")
    ot, _ = oracle.Cloud.load(tmp_path / "g_False_5.pwn")
    a, b = c2.arrays(stats=True), ot.arrays(stats=True)
    for k in ("points", "normals", "stats"):
        assert np.array_equal(_bits(a[k]), _bits(b[k])), k
    with pytest.raises(Exception):
        api.Cloud(gctx, 10).load(tmp_path / "o_True_1.pwn")                     # capacity error, not a crash


@pytest.mark.gpu
def test_merge_on_real_sensor_frames(gctx, oracle):
    """the two Kinect frames of tests/golden (PlaneEx_gui/test_images): aligned with the HIP aligner, fused with the merger"""
    from g2o_frontend_amd import api
    from test_gpu_parity import gpu_objects
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kinect_real_pair.npz"))
    rows, cols, K, conv, alig = case_params("vga")
    depths = [oracle.convert_16u_to_32f(z["ref_mm"]), oracle.convert_16u_to_32f(z["cur_mm"])]
    oracle.set_gaussians(True)
    try:
        cp = oracle.converter_params(K=K, **conv)
        o0, o1 = (oracle.convert(cp, d)[0] for d in depths)
    finally:
        oracle.set_gaussians(False)
    proj, converter, (g0, g1) = _gpu_convert(gctx, "vga", depths)
    _, _, aligner = gpu_objects(gctx, "vga")
    aligner.setReferenceCloud(g0); aligner.setCurrentCloud(g1)
    T = aligner.align()["T"]
    oscene = oracle.Cloud(); gscene = api.Cloud(gctx, 2 * rows * cols)
    oscene.add(o0, np.eye(4)); gscene.add(g0, np.eye(4)); oscene.add(o1, T); gscene.add(g1, T)
    merger = api.Merger(); merger.setDepthImageConverter(converter); merger.setImageSize(rows, cols)
    ok, ocol = oracle.merge(oscene, K, T, conv["min_distance"], conv["max_distance"], rows, cols)
    assert merger.merge(gscene, T) == ok and np.array_equal(merger.collapsedIndices(), ocol)
    assert ok < 0.9 * (len(o0) + len(o1))                                       # the overlap really fused (measured: 81.5 % left)
    _same_cloud(oscene, gscene)


@pytest.mark.gpu
def test_incremental_scene_harness(gctx, oracle):
    """The mapping loop of pwn_aligner.cpp:150-208 on a 5-frame synthetic stream (imageScale-4 configuration): render the scene
    into the current view, convert that rendering to a sub-scene, align the new frame against it, add the frame to the scene,
    merge.  The HIP side is teacher-forced with the oracle's poses, so after every frame the scene clouds, the rendered depth
    images and the sub-scene clouds are bit-identical; the HIP aligner's own relative motion matches the oracle's to 2e-4."""
    from g2o_frontend_amd import api, synth
    from test_gpu_parity import gpu_objects, oracle_params
    name = "small"
    rows, cols, K, conv, alig = case_params(name)
    n_frames = 5
    poses = synth.trajectory(3, n_frames)
    depths = [oracle.convert_16u_to_32f(synth.render_depth_mm(3, poses[k], rows, cols, K, hole_stream=k)) for k in range(n_frames)]
    cp, ap = oracle_params(oracle, name, accumulate_fp64=1)
    proj, converter, aligner = gpu_objects(gctx, name)
    merger = api.Merger(); merger.setDepthImageConverter(converter); merger.setImageSize(rows, cols)
    oscene = oracle.Cloud(); gscene = api.Cloud(gctx, n_frames * rows * cols)
    sceneT = np.eye(4, dtype=np.float32)
    I = np.eye(4, dtype=np.float32)
    oracle.set_gaussians(True)
    try:
        for k, d in enumerate(depths):
            oc, _, _ = oracle.convert(cp, d)
            gc = api.Cloud(gctx, rows * cols); converter.compute(gc, d, keep_stats=True, gaussians=True)
            if k > 0:
                # converter.projector()->setTransform(sceneT * sensorOffset); project(scaledIndexImage, scaledDepth, referenceScene->points())
                oidx, odep = oracle.project(K, sceneT, conv["min_distance"], conv["max_distance"], rows, cols, oscene.arrays()["points"])
                proj.setImageSize(rows, cols); proj.setTransform(sceneT)
                gidx, gdep = proj.project(gscene)
                assert np.array_equal(oidx, gidx) and np.array_equal(odep.view(np.uint32), gdep.view(np.uint32)), k
                osub, _, _ = oracle.convert(cp, odep)
                gsub = api.Cloud(gctx, rows * cols); converter.compute(gsub, gdep, keep_stats=True, gaussians=True)
                _same_cloud(osub, gsub)
                o = oracle.align(ap, osub, oc)
                aligner.setReferenceCloud(gsub); aligner.setCurrentCloud(gc); aligner.setInitialGuess(I)
                g = aligner.align()
                assert np.abs(g["T"] - o["T"]).max() < 2e-4, (k, np.abs(g["T"] - o["T"]).max())
                sceneT = oracle.iso_mul(sceneT, o["T"])                       # teacher-forced: the oracle's pose on both sides
                sceneT[3] = (0, 0, 0, 1)
                true = np.linalg.inv(poses[0]) @ poses[k]
                assert np.abs(sceneT[:3, 3] - true[:3, 3]).max() < 0.02, k   # the loop tracks the synthetic camera
            oscene.add(oc, sceneT); gscene.add(gc, sceneT)
            ok, ocol = oracle.merge(oscene, K, sceneT, conv["min_distance"], conv["max_distance"], rows, cols)
            assert merger.merge(gscene, sceneT) == ok and np.array_equal(merger.collapsedIndices(), ocol), k
            _same_cloud(oscene, gscene)
    finally:
        oracle.set_gaussians(False)
    assert len(oscene) < 0.6 * sum(int(((d >= conv["min_distance"]) & (d <= conv["max_distance"])).sum()) for d in depths)


@pytest.mark.gpu
def test_scene_edge_cases_and_error_paths(gctx, oracle):
    """empty clouds, missing Gaussians, capacity overflow, all-invalid frames: status codes, no crash, same behaviour as the oracle"""
    from g2o_frontend_amd import api
    from g2o_frontend_amd._lib import PwnHipError
    rows, cols, K, conv, depths, Ttrue, (o0, _) = _oracle_scene(oracle)
    proj, converter, (g0,) = _gpu_convert(gctx, "small", depths[:1])
    merger = api.Merger(); merger.setDepthImageConverter(converter); merger.setImageSize(rows, cols)
    # empty scene: merge and voxelize are no-ops
    empty = api.Cloud(gctx, 16)
    assert empty.size() == 0 and api.VoxelCalculator().compute(empty, 0.05) == 0
    # Merger::merge without Gaussians is refused (the reference would read an empty vector)
    nog = api.Cloud(gctx, rows * cols)
    converter.compute(nog, depths[0])
    with pytest.raises(PwnHipError):
        merger.merge(nog, np.eye(4))
    # Cloud::add beyond the destination capacity is a status code, and leaves the destination untouched
    small = api.Cloud(gctx, g0.size() + 10)
    small.add(g0, np.eye(4))
    before = small.arrays()["points"].copy()
    with pytest.raises(PwnHipError):
        small.add(g0, np.eye(4))
    assert small.size() == g0.size() and np.array_equal(small.arrays()["points"], before)
    with pytest.raises(PwnHipError):
        small.add(small, np.eye(4))
    # a frame without a single valid pixel: empty cloud, empty Gaussians, add / merge still fine
    blank = np.zeros((rows, cols), np.float32)
    oracle.set_gaussians(True)
    try:
        ob, _, _ = oracle.convert(oracle.converter_params(K=K, **conv), blank)
    finally:
        oracle.set_gaussians(False)
    gb = api.Cloud(gctx, rows * cols)
    converter.compute(gb, blank, keep_stats=True, gaussians=True)
    assert len(ob) == gb.size() == 0 and gb.numGaussians() == 0
    scene = api.Cloud(gctx, 2 * rows * cols); oscene = oracle.Cloud()
    scene.add(gb, Ttrue); oscene.add(ob, Ttrue)
    scene.add(g0, Ttrue); oscene.add(o0, Ttrue)
    ok, ocol = oracle.merge(oscene, K, Ttrue, conv["min_distance"], conv["max_distance"], rows, cols)
    assert merger.merge(scene, Ttrue) == ok and np.array_equal(merger.collapsedIndices(), ocol)
    _same_cloud(oscene, scene)
    # a view that sees nothing (looking away): every point keeps _collapsedIndices = -1, the cloud is unchanged
    away = np.array([[-1, 0, 0, 0], [0, 1, 0, 0], [0, 0, -1, -10], [0, 0, 0, 1]], np.float32)
    ok2, ocol2 = oracle.merge(oscene, K, away, conv["min_distance"], conv["max_distance"], rows, cols)
    assert merger.merge(scene, away) == ok2 == ok and np.all(ocol2 == -1) and np.array_equal(merger.collapsedIndices(), ocol2)
    _same_cloud(oscene, scene)
    # voxel grid: a NaN point is reported, not silently hashed
    bad = api.Cloud(gctx, 8)
    P = np.zeros((4, 4), np.float32); P[:, 3] = 1; P[2, 0] = np.nan
    bad.upload(P, np.zeros((4, 4), np.float32), np.zeros(4, np.float32), np.zeros((4, 16), np.float32), np.zeros((4, 16), np.float32))
    with pytest.raises(PwnHipError):
        api.VoxelCalculator().compute(bad, 0.01)
    # unreadable / foreign files
    with pytest.raises(PwnHipError):
        api.Cloud(gctx, 8).load("/nonexistent/file.pwn")


@pytest.mark.gpu
def test_scene_above_two_million_points_1280x960(oracle):
    """The mapping loop of pwn_aligner.cpp:205-208 at BASELINE configs[4]'s frame size: three 1280x960 frames added to one scene
    (3.6 M points: above the 2^21 the z-buffer words could index until round 4), Merger::merge from the last pose, then the voxel grid --
    every array bit-exact against the oracle, collapsed / kept indices equal.  The aligner itself keeps its 2^21-point word: handing it
    the scene is refused with PWN_HIP_ERR_CAPACITY."""
    from g2o_frontend_amd import api, synth
    from g2o_frontend_amd._lib import PwnHipError
    from test_gpu_parity import gpu_objects
    name = "k2"
    rows, cols, K, conv, _ = case_params(name)
    N = rows * cols
    poses = [np.eye(4, dtype=np.float32), synth.pair_pose(11).astype(np.float32), synth.pair_pose(12).astype(np.float32)]
    depths = [oracle.convert_16u_to_32f(synth.render_depth_mm(11, np.asarray(p, np.float64), rows, cols, K)) for p in poses]
    oracle.set_gaussians(True)
    try:
        cp = oracle.converter_params(K=K, **conv)
        oclouds = [oracle.convert(cp, d)[0] for d in depths]
    finally:
        oracle.set_gaussians(False)
    ctx = api.Context(0, rows, cols, 2, omega_storage="exact9")
    proj, converter, aligner = gpu_objects(ctx, name)
    gclouds = []
    for d in depths:
        c = api.Cloud(ctx, N)
        converter.compute(c, d, keep_stats=True, gaussians=True)
        gclouds.append(c)
    oscene = oracle.Cloud(); gscene = api.Cloud(ctx, 3 * N)
    for oc, gc, T in zip(oclouds, gclouds, poses):
        oscene.add(oc, T); gscene.add(gc, T)
    n0 = len(oscene)
    assert n0 == gscene.size() and n0 > (1 << 21), n0
    _same_cloud(oscene, gscene)
    # the projection of the large scene as an image (stand-alone projector: the same 64-bit words as the merger's)
    oi, od = oracle.project(K, poses[1], conv["min_distance"], conv["max_distance"], rows, cols, oscene.arrays()["points"])
    proj.setTransform(poses[1]); proj.setImageSize(rows, cols)
    gi, gd = proj.project(gscene)
    assert np.array_equal(oi, gi) and np.array_equal(od.view(np.uint32), gd.view(np.uint32)) and gi.max() >= (1 << 21)
    with pytest.raises(PwnHipError) as e:
        aligner.setReferenceCloud(gscene); aligner.setCurrentCloud(gclouds[0]); aligner.align()
    assert e.value.code == 6
    merger = api.Merger(); merger.setDepthImageConverter(converter); merger.setImageSize(rows, cols)
    ok, ocol = oracle.merge(oscene, K, poses[2], conv["min_distance"], conv["max_distance"], rows, cols)
    gk = merger.merge(gscene, poses[2])
    assert gk == ok and np.array_equal(merger.collapsedIndices(), ocol)
    assert ((ocol >= 0) & (ocol != np.arange(len(ocol)))).sum() > 100000 and ocol.max() >= (1 << 21)      # targets beyond the old index field took part
    _same_cloud(oscene, gscene)
    vox = api.VoxelCalculator()
    ok, okept = oracle.voxelize(oscene, 0.02, literal=False)
    gk = vox.compute(gscene, 0.02)
    assert gk == ok and np.array_equal(vox.keptIndices(), okept)
    _same_cloud(oscene, gscene, gauss=False)
    print(f"scene of {n0} points: merge -> {len(ocol) - int(((ocol >= 0) & (ocol != np.arange(len(ocol)))).sum())}, voxel grid (2 cm) -> {ok}")
    ctx.close()
