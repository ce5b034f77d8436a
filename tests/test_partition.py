"""SURVEY.md 8(e)'s processPartition mode: one `current` cloud against the cached clouds of the other partition, the clouds sharded over the
GPUs of a node and `current` replicated by one broadcast (pwn_tracker/pwn_closer.cpp:85-111).

  * the flat form of a cloud (pwn_hip_cloud_export / _import) is the cloud, bit for bit: arrays, index image, projection shortcuts, and
    therefore every alignment it takes part in -- through device and host buffers, into the same and into another context;
  * pwn_hip_match_batch_records: the 72-float records written on the device carry what pwn_hip_match_batch returns on the host;
  * bench.py --mode partition: the line of a one-rank run, and the same run with broadcast + all-gather forced through RCCL (world size 1,
    rank 0 matching against the REPLICA that travelled through export / broadcast / import): records equal CRC by CRC.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import case_params, make_depth_pair

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype.itemsize == 4 else a


def _same_cloud(a, b):
    A, B = a.arrays(), b.arrays()
    assert a.size() == b.size()
    for k in A:
        assert np.array_equal(_bits(A[k]), _bits(B[k])), k


@pytest.mark.parametrize("storage", ["exact9", "sym6"])
def test_flat_cloud_round_trip_is_the_cloud_bit_for_bit(storage):
    from g2o_frontend_amd import api
    from test_gpu_parity import gpu_objects
    name = "small"
    rows, cols, K, conv, alig = case_params(name)
    ref, cur, _, _, _ = make_depth_pair(name, 3)
    ctx = api.Context(0, rows, cols, 4, omega_storage=storage)
    _, converter, aligner = gpu_objects(ctx, name)
    N = rows * cols
    gref, gcur = api.Cloud(ctx, N), api.Cloud(ctx, N)
    converter.compute(gref, ref); converter.compute(gcur, cur)
    bound = api.Cloud.flatBound(N, storage, N)
    assert gref.flatSize() <= bound and gref.flatSize() % 256 == 0
    # device buffer, same context
    flat = ctx.upload(np.zeros(bound, np.uint8))              # a device buffer (the library's own allocator: no torch in this process)
    used = gref.exportFlat(flat)
    assert used == gref.flatSize()
    rep = api.Cloud(ctx, N)
    rep.importFlat(flat)
    _same_cloud(gref, rep)
    # host buffer (numpy), exact size
    host = np.zeros(gcur.flatSize(), np.uint8)
    assert gcur.exportFlat(host) == host.size
    rep_cur = api.Cloud(ctx, N)
    rep_cur.importFlat(host)
    _same_cloud(gcur, rep_cur)
    # the bytes themselves: device and host exports of one cloud are the same bytes
    host2 = np.zeros(used, np.uint8); gref.exportFlat(host2)
    assert np.array_equal(flat.numpy()[:used], host2)
    # alignments: replica in either role, and in both, are bitwise the original's (the replica carries the converter's index image, so it takes
    # the same projection shortcuts)
    aligner.setReferenceCloud(gref); aligner.setCurrentCloud(gcur)
    base = aligner.align()
    for r, c in ((rep, gcur), (gref, rep_cur), (rep, rep_cur)):
        aligner.setReferenceCloud(r); aligner.setCurrentCloud(c)
        g = aligner.align()
        for k in ("T", "chi2", "C", "K", "iter_inliers"):
            assert np.array_equal(_bits(g[k]), _bits(base[k])), k
    batch = aligner.alignBatch([gref, rep, rep], [gcur, gcur, rep_cur])
    for g in batch:
        assert np.array_equal(_bits(g["T"]), _bits(base["T"])) and np.array_equal(_bits(g["chi2"]), _bits(base["chi2"]))
    # another context (what another rank's process does with the broadcast buffer), larger capacity than the cloud needs
    ctx2 = api.Context(0, rows, cols, 2, omega_storage=storage)
    _, _, aligner2 = gpu_objects(ctx2, name)
    far_ref, far_cur = api.Cloud(ctx2, 2 * N), api.Cloud(ctx2, N)
    far_ref.importFlat(flat); far_cur.importFlat(host)
    _same_cloud(gref, far_ref); _same_cloud(gcur, far_cur)
    aligner2.setReferenceCloud(far_ref); aligner2.setCurrentCloud(far_cur)
    g = aligner2.align()
    assert np.array_equal(_bits(g["T"]), _bits(base["T"])) and np.array_equal(_bits(g["chi2"]), _bits(base["chi2"]))
    # errors: short buffers, foreign bytes, the other omega storage, a cloud that is too small
    with pytest.raises(api.PwnHipError) as e:
        gref.exportFlat(np.zeros(used - 256, np.uint8))
    assert e.value.code == 6
    with pytest.raises(api.PwnHipError):
        rep.importFlat(np.zeros(4096, np.uint8))
    with pytest.raises(api.PwnHipError):
        rep.importFlat(host2[: used - 256])
    other = "sym6" if storage == "exact9" else "exact9"
    ctx3 = api.Context(0, rows, cols, 1, omega_storage=other)
    with pytest.raises(api.PwnHipError):
        api.Cloud(ctx3, N).importFlat(host2)
    with pytest.raises(api.PwnHipError) as e:
        api.Cloud(ctx2, 16).importFlat(host2)
    assert e.value.code == 6
    _same_cloud(gref, rep)                       # failed imports above went to other clouds; this one is untouched
    flat.free()
    for c in (ctx3, ctx2, ctx):
        c.close()


def test_flat_cloud_edge_cases_empty_cloud_and_empty_batch():
    """a frame without a single valid pixel gives an empty cloud: its flat form is header + index image, the replica is empty too and aligns like the
    original (zero correspondences, pose = guess); a match batch of zero pairs is a no-op"""
    from g2o_frontend_amd import api
    from test_gpu_parity import gpu_objects
    name = "small"
    rows, cols, K, conv, alig = case_params(name)
    ref, _, _, _, _ = make_depth_pair(name, 5)
    ctx = api.Context(0, rows, cols, 4)
    _, converter, aligner = gpu_objects(ctx, name)
    N = rows * cols
    empty, full = api.Cloud(ctx, N), api.Cloud(ctx, N)
    converter.compute(empty, np.zeros((rows, cols), np.float32)); converter.compute(full, ref)
    assert empty.size() == 0
    buf = np.zeros(empty.flatSize(), np.uint8)
    assert empty.exportFlat(buf) == buf.size and buf.size == 256 + ((N * 4 + 255) // 256) * 256      # header + the (all -1) index image
    rep = api.Cloud(ctx, N)
    converter.compute(rep, ref)                      # the destination held something else before
    rep.importFlat(buf)
    assert rep.size() == 0
    aligner.setReferenceCloud(full); aligner.setCurrentCloud(empty)
    a = aligner.align()
    aligner.setCurrentCloud(rep)
    b = aligner.align()
    assert np.array_equal(_bits(a["T"]), _bits(b["T"])) and np.array_equal(a["C"], b["C"]) and int(a["C"].sum()) == 0
    alproj = api.PinholePointProjector(); alproj.setMinDistance(alig["min_distance"]); alproj.setMaxDistance(alig["max_distance"])
    aligner.setProjector(alproj)
    matcher = api.PwnMatcherBase(aligner, converter); matcher.setScale(1)
    Km = np.array([[K[0], 0, K[2]], [0, K[1], K[3]], [0, 0, 1]], np.float32); I = np.eye(4, dtype=np.float32)
    rec = np.full((1, api.MATCH_RECORD_FLOATS), -5.0, np.float32)
    out = matcher.matchCloudsBatchRecords([], [], I, I, Km, rows, cols, rec)
    assert len(out[0]) == 0 and (rec == -5.0).all()
    ctx.close()


def test_flat_cloud_of_an_uploaded_cloud_carries_its_normal_information_planes(oracle):
    """clouds that did not come from the converter hold full normal information matrices instead of a class (pwn_hip_cloud_upload) and no index image"""
    from g2o_frontend_amd import api
    from test_gpu_parity import oracle_params, upload
    name = "small"
    rows, cols, _, _, _ = case_params(name)
    ref, _, _, _, _ = make_depth_pair(name, 2)
    cp, _ = oracle_params(oracle, name)
    oref, _, _ = oracle.convert(cp, ref)
    ctx = api.Context(0, rows, cols, 1)
    up = upload(ctx, oref)
    buf = np.zeros(up.flatSize(), np.uint8)
    up.exportFlat(buf)
    rep = api.Cloud(ctx, rows * cols)
    rep.importFlat(buf)
    _same_cloud(up, rep)
    ctx.close()


def test_match_batch_records_carry_the_matcher_results():
    from g2o_frontend_amd import api
    from test_gpu_parity import gpu_objects
    sys.path.insert(0, ROOT)
    import bench
    name = "small"
    rows, cols, K, conv, alig = case_params(name)
    from g2o_frontend_amd import synth
    ctx = api.Context(0, rows, cols, 16)
    _, converter, aligner = gpu_objects(ctx, name)
    alproj = api.PinholePointProjector(); alproj.setMinDistance(alig["min_distance"]); alproj.setMaxDistance(alig["max_distance"])
    aligner.setProjector(alproj)
    matcher = api.PwnMatcherBase(aligner, converter); matcher.setScale(1)
    Km = np.array([[K[0], 0, K[2]], [0, K[1], K[3]], [0, 0, 1]], np.float32); I = np.eye(4, dtype=np.float32)
    n = 9
    ids = list(range(n))
    cur_mm = synth.render_depth_mm(bench.PARTITION_SCENE, np.eye(4), rows, cols, K, hole_stream=0)
    others_mm = [bench._render_job(j) for j in bench.partition_jobs(ids, rows, cols, K)]
    N = rows * cols
    current = api.Cloud(ctx, N)
    others = [api.Cloud(ctx, N) for _ in range(n)]
    converter.computeBatch([current] + others, [cur_mm] + others_mm, raw_scale=0.001)
    guesses = bench.partition_guesses(ids)
    base = matcher.matchCloudsBatch([current] * n, others, I, I, Km, rows, cols, guesses)
    rec_dev = ctx.upload(np.full((n, api.MATCH_RECORD_FLOATS), -7.0, np.float32))
    pid = np.arange(100, 100 + n, dtype=np.int32)
    res, sc = matcher.matchCloudsBatchRecords([current] * n, others, I, I, Km, rows, cols, rec_dev, guesses, pair_ids=pid)
    rec = rec_dev.numpy()
    rec_host = np.full((n, api.MATCH_RECORD_FLOATS), -7.0, np.float32)
    matcher.matchCloudsBatchRecords([current] * n, others, I, I, Km, rows, cols, rec_host, guesses, pair_ids=pid, want_results=False)
    assert np.array_equal(_bits(rec), _bits(rec_host))
    for i in range(n):
        b = base[i]
        assert np.array_equal(_bits(rec[i, :16]), _bits(np.ascontiguousarray(b["align"]["T"].T.reshape(-1))))
        assert rec[i, 17] == b["cloud_inliers"] and rec[i, 18] == 10 and rec[i, 19] == 100 + i
        assert np.array_equal(_bits(rec[i, 20:30]), _bits(b["align"]["chi2"][:10]))
        assert (rec[i, 64], rec[i, 65], rec[i, 66]) == (b["image_nonZeros"], b["image_outliers"], b["image_inliers"])
        assert np.array_equal(_bits(rec[i, 67:68]), _bits(np.float32([b["image_reprojectionDistance"]])))
        assert np.all(rec[i, 68:] == 0)
        assert (sc[i].image_non_zeros, sc[i].image_inliers) == (b["image_nonZeros"], b["image_inliers"])
        assert np.array_equal(_bits(res["T"][i]), _bits(rec[i, :16]))
        # the alignments found the relative pose of the views (same scene, <= 5 cm / 2.3 deg apart, guess 1 cm / 0.5 deg off)
        assert np.abs(b["align"]["T"][:3, 3] - synth.pair_pose(bench.PARTITION_POSE0 + i)[:3, 3]).max() < 2e-2, i
        assert b["image_nonZeros"] > N // 2
    # include/pwn_hip.h: results and scores may be NULL independently -- scores without results are filled in all the same
    import ctypes as C
    from g2o_frontend_amd._lib import MatchResult
    refs, curs, g, _ = matcher.matchHandles([current] * n, others, guesses)
    only = (MatchResult * n)()
    rec3 = np.zeros((n, api.MATCH_RECORD_FLOATS), np.float32)
    p = aligner.params()
    ctx.check(ctx._L.pwn_hip_match_batch_records(ctx.h, C.byref(p), n, refs, curs, g.ctypes.data_as(C.c_void_p), 50.0, None, 100, None, only, rec3.ctypes.data_as(C.c_void_p)))
    assert np.array_equal(_bits(rec3), _bits(rec))
    for i in range(n):
        assert (only[i].image_non_zeros, only[i].image_inliers, only[i].image_outliers) == (sc[i].image_non_zeros, sc[i].image_inliers, sc[i].image_outliers)
        assert np.float32(only[i].image_reprojection_distance).view(np.uint32) == np.float32(sc[i].image_reprojection_distance).view(np.uint32)
    # a records buffer that is too small or of the wrong type never reaches the library
    with pytest.raises(ValueError):
        matcher.matchCloudsBatchRecords([current] * n, others, I, I, Km, rows, cols, ctx.upload(np.zeros((n, 64), np.float32)), guesses)
    with pytest.raises(ValueError):
        matcher.matchCloudsBatchRecords([current] * n, others, I, I, Km, rows, cols, np.empty((n, 72), np.float64), guesses)
    with pytest.raises(ValueError):
        aligner.alignBatchRecords([current] * n, others, np.empty((n - 1, 64), np.float32))
    ctx.close()


def test_bench_partition_mode_original_and_replica_give_the_same_records(tmp_path):
    """bench.py --mode partition on one GPU: (1) plain -- every pair matched against the converted `current` cloud; (2) broadcast and all-gather
    forced through RCCL at world size 1 -- rank 0 matches against the replica that went through export / broadcast / import.  Both lines carry the
    roofline and the closer's acceptance count; the assembled 288-byte records are equal, CRC by CRC (run 1 writes the digests into a scratch
    directory, run 2 is checked against them)."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    crc_dir = str(tmp_path)
    # bench.py reads its CRC directory from the module constant: run it through a tiny driver that redirects it
    driver = tmp_path / "run_bench.py"
    driver.write_text("import sys; sys.path.insert(0, %r); import bench; bench.CRC_DIR = %r; sys.argv[0] = %r; bench.main()\n"
                      % (ROOT, crc_dir, os.path.join(ROOT, "bench.py")))

    def run(env, args):
        e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "PWN_BENCH_FORCE_DIST")}
        e.update(env)
        out = subprocess.run([sys.executable, str(driver), "--gpus", "1", "--mode", "partition", "--pairs", "40", "--steps", "2", "--warmup", "1",
                              "--no-cpu-baseline", "--render-workers", "1"] + args, env=e, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        return json.loads(out.stdout.strip().splitlines()[-1])

    plain = run({}, ["--write-records-crc"])
    assert plain["config"]["mode"] == "partition" and plain["value"] > 0 and plain["gather"]["records"] == 40 and plain["gather"]["record_bytes"] == 288
    assert plain["gather"]["records_equal_local"] is True
    assert plain["gather"]["records_vs_single_gpu_run"]["checked"] == 40 and plain["gather"]["records_vs_single_gpu_run"]["equal"] is True
    assert plain["partition"]["accepted_by_closer_thresholds_rank0"] >= 30 and plain["partition"]["max_translation_error_m_rank0"] < 1e-2
    assert 0.2 < plain["roofline"]["frac"] < 1.0 and plain["roofline"]["projections_per_pair"] == 10.0      # non-identity guesses: every reference projection runs
    forced = run(dict(PWN_BENCH_FORCE_DIST="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY="0"), [])
    g = forced["gather"]
    assert g["backend"].startswith("nccl") and g["forced"] is True and forced["partition"]["replica_roundtrip_on_rank0"] is True
    assert g["records_vs_single_gpu_run"]["checked"] == 40 and g["records_vs_single_gpu_run"]["equal"] is True, g
    assert g["records_vs_single_gpu_run"]["file_is_for_these_kernels"] is True
    m = forced["multi_gpu"]
    assert len(m["per_rank_ms_per_step"]) == 1 and m["per_rank_ms_per_step"][0] > 0
    assert m["collectives_alone"]["broadcast_ms"] > 0 and m["collectives_alone"]["gather_ms"] > 0
    assert 10e6 < m["collectives_alone"]["flat_cloud_bytes"] <= m["collectives_alone"]["broadcast_bytes"]
    assert set(m["per_rank_stage_ms_per_step"]) >= {"corr_linearize", "project_ref", "solve", "match_score"}


@pytest.mark.parametrize("mode", ["pipelined", "serial"])
def test_native_rccl_partition_app_matches_the_python_mirror(tmp_path, mode):
    """tools/pwn_hip_partition_app.cpp -- the processPartition flow of INTEGRATION.md in native code: C++ mirror over the C-ABI, RCCL for the broadcast
    of the flat `current` cloud and the all-gather of the 288-byte records (world size 1 here: one GPU; the ranks of a larger run are forked before
    any GPU call and meet through an ncclUniqueId passed over pipes).  Its printed records must equal, digit for digit, what the Python mirror's
    matchCloudsBatch gives for the same frames and guesses on the converted `current` cloud."""
    sys.path.insert(0, ROOT)
    import bench
    from g2o_frontend_amd import api, build, synth
    build.build_tools()
    exe = os.path.join(ROOT, "tools", "pwn_hip_partition_app")
    if not os.path.exists(exe):
        pytest.skip("no RCCL headers: the app was not built")
    rows, cols, K = 480, 640, synth.K_VGA
    n = 6
    ids = list(range(n))
    frames = [synth.render_depth_mm(bench.PARTITION_SCENE, np.eye(4), rows, cols, K, hole_stream=0)] + [bench._render_job(j) for j in bench.partition_jobs(ids, rows, cols, K)]
    names = []
    for k, f in enumerate(frames):
        fn = tmp_path / f"f{k}.pgm"
        with open(fn, "wb") as fh:
            fh.write(b"P5\n%d %d\n65535\n" % (cols, rows)); fh.write(f.astype(">u2").tobytes())
        names.append(str(fn))
    (tmp_path / "frames.txt").write_text("\n".join(names) + "\n")
    guesses = [np.asarray(g, np.float64).astype(np.float32) for g in bench.partition_guesses(ids)]
    (tmp_path / "guesses.txt").write_text("\n".join(" ".join("%.9g" % v for v in g.T.reshape(-1)) for g in guesses) + "\n")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([exe, str(tmp_path / "frames.txt"), "1", "3", str(tmp_path / "guesses.txt")] + (["serial"] if mode == "serial" else []),
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, (out.stdout[-500:], out.stderr[-1500:])
    lines = [l.split() for l in out.stdout.splitlines()]
    head = [l for l in lines if l and l[0] == "keyframes"][0]
    assert (int(head[1]), int(head[3])) == (n, 1) and float(head[7]) > 0 and int(head[9]) > 10e6 and head[13] == mode
    # pipelined: only the bytes written travel; serial: the buffer's bound
    assert (int(head[11]) == int(head[9])) if mode == "pipelined" else (int(head[11]) >= int(head[9]))
    recs = [l for l in lines if l and l[0] == "keyframe"]
    assert len(recs) == n
    # the same flow with the Python mirror: sym6 clouds, scale 1, bench.py's VGA tables
    ctx = api.Context(0, rows, cols, 16, omega_storage="sym6")
    Kc, conv, alig = bench.conf(rows, cols)
    converter, aligner = bench.build_objects(ctx, rows, cols, Kc, conv, alig)
    alproj = api.PinholePointProjector(); alproj.setMinDistance(alig["min_distance"]); alproj.setMaxDistance(alig["max_distance"])
    aligner.setProjector(alproj)
    matcher = api.PwnMatcherBase(aligner, converter); matcher.setScale(1)
    Km = np.array([[K[0], 0, K[2]], [0, K[1], K[3]], [0, 0, 1]], np.float32); I = np.eye(4, dtype=np.float32)
    clouds = [api.Cloud(ctx, rows * cols) for _ in range(n + 1)]
    converter.computeBatch(clouds, frames, raw_scale=0.001)
    base = matcher.matchCloudsBatch([clouds[0]] * n, clouds[1:], I, I, Km, rows, cols, guesses)
    acc = api.PwnCloserAcceptance()
    for k, (l, b) in enumerate(zip(recs, base)):
        v = np.array(l[1:], np.float64)
        assert int(v[0]) == k and bool(v[1]) == acc.accept(b)
        assert (int(v[2]), int(v[3]), int(v[4]), int(v[5])) == (b["cloud_inliers"], b["image_nonZeros"], b["image_outliers"], b["image_inliers"]), (k, v[:7], b)
        assert np.float32(v[6]) == np.float32(b["image_reprojectionDistance"])
        assert np.array_equal(v[7:23].astype(np.float32).reshape(4, 4).T, b["align"]["T"]), k
        assert np.array_equal(v[23:33].astype(np.float32), b["align"]["chi2"][:10]), k
    ctx.close()


def test_partition_workload_against_the_oracle(oracle):
    """The workload of bench.py --mode partition itself (VGA, sym6 clouds, one `current` cloud as the reference of every pair, odometry guesses with the z
    translation zeroed) against the oracle on sampled keyframes: converter clouds (sym6 comparison), every iteration of the oracle's ten-iteration trace
    re-run from the oracle's iterate (chi2 1e-5 vs the fp64-accumulated sums, K_i / C_i / inliers_i exact), the batch's own first-iteration counters exact
    and its final pose within 1e-5 of the oracle's free run, and the depth-agreement score: exact against the oracle's scoring of the GPU's own finder
    images, within a few pixels of the oracle's own chain."""
    sys.path.insert(0, ROOT)
    import bench
    from g2o_frontend_amd import api, synth
    from test_gpu_parity import _check_teacher_forced
    from test_omega_sym6 import compare_clouds_sym6
    rows, cols, K = 480, 640, synth.K_VGA
    n = 8
    ids = list(range(n))
    cur_mm = synth.render_depth_mm(bench.PARTITION_SCENE, np.eye(4), rows, cols, K, hole_stream=0)
    others_mm = [bench._render_job(j) for j in bench.partition_jobs(ids, rows, cols, K)]
    ctx = api.Context(0, rows, cols, 16, omega_storage="sym6")
    Kc, conv, alig = bench.conf(rows, cols)
    converter, aligner = bench.build_objects(ctx, rows, cols, Kc, conv, alig)
    alproj = api.PinholePointProjector(); alproj.setMinDistance(alig["min_distance"]); alproj.setMaxDistance(alig["max_distance"])
    aligner.setProjector(alproj)
    matcher = api.PwnMatcherBase(aligner, converter); matcher.setScale(1)
    Km = np.array([[K[0], 0, K[2]], [0, K[1], K[3]], [0, 0, 1]], np.float32); I = np.eye(4, dtype=np.float32)
    clouds = [api.Cloud(ctx, rows * cols) for _ in range(n + 1)]
    converter.computeBatch(clouds, [cur_mm] + others_mm, raw_scale=0.001)
    guesses = bench.partition_guesses(ids)
    rec = np.zeros((n, api.MATCH_RECORD_FLOATS), np.float32)
    res, sc = matcher.matchCloudsBatchRecords([clouds[0]] * n, clouds[1:], I, I, Km, rows, cols, rec, guesses)
    cp = oracle.converter_params(K=K, **conv)
    ocur, _, _ = oracle.convert(cp, oracle.convert_16u_to_32f(cur_mm))
    compare_clouds_sym6(ocur.arrays(), clouds[0].arrays())
    worst_chi2 = worst_pose = 0.0
    for i in (0, 3, 7):
        ooth, _, _ = oracle.convert(cp, oracle.convert_16u_to_32f(others_mm[i]))
        compare_clouds_sym6(ooth.arrays(), clouds[1 + i].arrays())
        g = np.asarray(guesses[i], np.float64).astype(np.float32); g[2, 3] = 0; g[3] = (0, 0, 0, 1)          # matchClouds' conditioning (pwn_matcher_base.cpp:114)
        ap = oracle.aligner_params(rows, cols, K=K, initial_guess=g, accumulate_fp64=1, **alig)
        o = oracle.align(ap, ocur, ooth, images=True)
        it0 = o["iterations"][0]
        assert (int(res["iter_candidates"][i][0]), int(res["iter_correspondences"][i][0]), int(res["iter_inliers"][i][0])) == (it0["K"], it0["C"], it0["inliers"]), i
        d = float(np.abs(res["T"][i].reshape(4, 4).T - o["T"]).max()); worst_pose = max(worst_pose, d)
        assert d <= 1e-5, (i, d)
        # teacher-forced: the aligner as matchClouds left it configured (projector, finder size), this pair's clouds
        aligner.setReferenceCloud(clouds[0]); aligner.setCurrentCloud(clouds[1 + i])
        worst_chi2 = max(worst_chi2, _check_teacher_forced(aligner, o))
        # score: the oracle's scoring of the GPU's own finder images of this pair is what the record carries
        aligner.setInitialGuess(g)
        aligner.align(images=True)
        f = aligner.correspondenceFinder()
        ex = oracle.match_score(f.referenceDepthImage(), f.currentDepthImage(), 50.0)
        assert (int(rec[i, 64]), int(rec[i, 65]), int(rec[i, 66])) == (ex["image_nonZeros"], ex["image_outliers"], ex["image_inliers"]), (i, rec[i, 64:68], ex)
        assert abs(float(rec[i, 67]) - ex["image_reprojectionDistance"]) <= 1e-3 * ex["image_reprojectionDistance"] + 1e-6
        os_ = oracle.match_score(o["ref_depth"], o["cur_depth"], 50.0)
        assert abs(int(rec[i, 64]) - os_["image_nonZeros"]) <= 8 and abs(int(rec[i, 66]) - os_["image_inliers"]) <= 8, (i, rec[i, 64:68], os_)
    print(f"partition workload, 3 of {n} keyframes vs oracle: worst teacher-forced chi2 rel diff {worst_chi2:.1e}, worst |T - T_oracle| {worst_pose:.1e}")
    ctx.close()


def test_flat_clouds_and_match_records_on_disturbed_inputs():
    """The same two properties on inputs the smooth room does not produce (tests/test_gpu_fuzz.py: sensor noise, dropouts, holes, exactly planar and
    constant patches, out-of-range rows -- clouds with non-finite information matrices and zero normals among them), random destination capacities and
    buffer kinds, both omega storages: replica == original in every array, and matching against the replica gives the original's records bit for bit."""
    from g2o_frontend_amd import api
    from test_gpu_fuzz import disturbed_pair
    from test_gpu_parity import gpu_objects
    name = "small"
    rows, cols, K, conv, alig = case_params(name)
    N = rows * cols
    rng = np.random.default_rng(77)
    for storage in ("exact9", "sym6"):
        ctx = api.Context(0, rows, cols, 16, omega_storage=storage)
        _, converter, aligner = gpu_objects(ctx, name)
        alproj = api.PinholePointProjector(); alproj.setMinDistance(alig["min_distance"]); alproj.setMaxDistance(alig["max_distance"])
        aligner.setProjector(alproj)
        matcher = api.PwnMatcherBase(aligner, converter); matcher.setScale(1)
        Km = np.array([[K[0], 0, K[2]], [0, K[1], K[3]], [0, 0, 1]], np.float32); I = np.eye(4, dtype=np.float32)
        frames = []
        for seed in range(6):
            a, b, _ = disturbed_pair(seed, name)
            frames += [a, b]
        clouds = [api.Cloud(ctx, N) for _ in frames]
        converter.computeBatch(clouds, frames, raw_scale=0.001)
        replicas = []
        for c in clouds:
            cap = int(rng.integers(max(1, c.size()), 2 * N))
            r = api.Cloud(ctx, cap)
            if rng.random() < 0.5:
                buf = np.zeros(c.flatSize() + int(rng.integers(0, 3)) * 256, np.uint8)       # exact or larger host buffer
                c.exportFlat(buf); r.importFlat(buf)
            else:
                buf = ctx.upload(np.zeros(api.Cloud.flatBound(N, storage, N), np.uint8))
                c.exportFlat(buf); r.importFlat(buf); buf.free()
            _same_cloud(c, r)
            replicas.append(r)
        n = len(frames) // 2
        guesses = []
        for i in range(n):
            g = np.eye(4); g[:3, 3] = rng.uniform(-0.01, 0.01, 3)
            guesses.append(g)
        rec_a = np.zeros((n, api.MATCH_RECORD_FLOATS), np.float32); rec_b = np.zeros_like(rec_a)
        matcher.matchCloudsBatchRecords(clouds[0::2], clouds[1::2], I, I, Km, rows, cols, rec_a, guesses, want_results=False)
        matcher.matchCloudsBatchRecords(replicas[0::2], replicas[1::2], I, I, Km, rows, cols, rec_b, guesses, want_results=False)
        assert np.array_equal(_bits(rec_a), _bits(rec_b)), storage
        assert (rec_a[:, 18] == 10).all() and (rec_a[:, 64] > 0).all()
        ctx.close()


def test_flat_cloud_at_1280x960():
    """BASELINE configs[4]'s frame size: a 1.2 M-point cloud (sym6: 64 MB flat) through a device buffer into another context; arrays equal, and the
    replica aligns like the original (one pair, identity guess)."""
    from g2o_frontend_amd import api, synth
    from test_gpu_parity import gpu_objects
    name = "k2"
    rows, cols, K, conv, alig = case_params(name)
    ref_mm, cur_mm, _ = synth.make_pair(11, rows, cols, K)
    ctx = api.Context(0, rows, cols, 2, omega_storage="sym6")
    _, converter, aligner = gpu_objects(ctx, name)
    N = rows * cols
    a, b = api.Cloud(ctx, N), api.Cloud(ctx, N)
    converter.computeBatch([a, b], [ref_mm, cur_mm], raw_scale=0.001)
    assert a.size() > 1_000_000
    flat = ctx.upload(np.zeros(api.Cloud.flatBound(N, "sym6", N), np.uint8))
    used = a.exportFlat(flat)
    assert 60e6 < used <= flat.nbytes
    ctx2 = api.Context(0, rows, cols, 2, omega_storage="sym6")
    _, _, aligner2 = gpu_objects(ctx2, name)
    ra, rb = api.Cloud(ctx2, N), api.Cloud(ctx2, N)
    ra.importFlat(flat)
    b.exportFlat(flat); rb.importFlat(flat)
    _same_cloud(a, ra); _same_cloud(b, rb)
    aligner.setReferenceCloud(a); aligner.setCurrentCloud(b)
    base = aligner.align()
    aligner2.setReferenceCloud(ra); aligner2.setCurrentCloud(rb)
    g = aligner2.align()
    for k in ("T", "chi2", "C", "K", "iter_inliers"):
        assert np.array_equal(_bits(g[k]), _bits(base[k])), k
    flat.free(); ctx2.close(); ctx.close()


def test_native_partition_app_does_not_hang_when_a_rank_fails(tmp_path):
    """ranks = 2 on a box with ONE GPU: rank 1 finds no device of its own and exits before the communicator exists, rank 0 waits for it in
    ncclCommInitRank.  The parent reaps the failed rank and terminates the other instead of waiting for ever (advisor finding, round 5); the
    run ends with the failed rank's code within seconds.  Same for a keyframe file only rank 1 reads that does not exist."""
    import time
    sys.path.insert(0, ROOT)
    import bench
    from g2o_frontend_amd import api, build, synth
    build.build_tools()
    exe = os.path.join(ROOT, "tools", "pwn_hip_partition_app")
    if not os.path.exists(exe):
        pytest.skip("no RCCL headers: the app was not built")
    if api.device_count() != 1:
        pytest.skip("needs a box with exactly one GPU")
    rows, cols, K = 120, 160, synth.scaled_K(synth.K_VGA, 4)
    f = synth.render_depth_mm(bench.PARTITION_SCENE, np.eye(4), rows, cols, K, hole_stream=0)
    names = []
    for k in range(3):
        fn = tmp_path / f"f{k}.pgm"
        with open(fn, "wb") as fh:
            fh.write(b"P5\n%d %d\n65535\n" % (cols, rows)); fh.write(f.astype(">u2").tobytes())
        names.append(str(fn))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PWN_PARTITION_TIMEOUT_S="60")
    (tmp_path / "frames.txt").write_text("\n".join(names) + "\n")
    t0 = time.time()
    out = subprocess.run([exe, str(tmp_path / "frames.txt"), "2", "1"], capture_output=True, text=True, timeout=120, env=env)
    assert out.returncode == 1 and time.time() - t0 < 45, (out.returncode, out.stderr[-800:])
    assert "one GPU per rank" in out.stderr
    (tmp_path / "frames2.txt").write_text("\n".join(names[:2] + [str(tmp_path / "missing.pgm")]) + "\n")
    t0 = time.time()
    out = subprocess.run([exe, str(tmp_path / "frames2.txt"), "2", "1"], capture_output=True, text=True, timeout=120, env=env)
    assert out.returncode == 1 and time.time() - t0 < 45 and "cannot read" in out.stderr, (out.returncode, out.stderr[-800:])
