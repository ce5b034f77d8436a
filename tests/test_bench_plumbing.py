"""bench.py's host-side plumbing on the CPU: the N-rank launcher (`python bench.py --gpus N` starts N ranks itself), sharding,
the result records in the real pwn_hip_align_result layout, the gloo all-gather, and the helpers of the extra bench lines."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def test_gpus_flag_launches_that_many_ranks_and_gathers_every_record():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run-cpu", "--pairs", "8"], capture_output=True, text=True, timeout=300,
                         env=_clean_env())
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout                       # ONE JSON line, from rank 0 only
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["records"] == 16 and j["records_ok"] and j["max_rank_seen"] == 1


def test_single_rank_dry_run_and_world_size_mismatch_is_refused():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--dry-run-cpu", "--pairs", "5"], capture_output=True, text=True, timeout=300, env=_clean_env())
    assert out.returncode == 0, out.stderr[-2000:]
    j = json.loads(out.stdout.strip().splitlines()[-1])
    assert j["n_gpus"] == 1 and j["records"] == 5 and j["records_ok"]
    env = _clean_env(); env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    bad = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run-cpu"], capture_output=True, text=True, timeout=300, env=env)
    assert bad.returncode == 2 and "refusing" in bad.stderr


def test_a_failing_rank_fails_the_launcher():
    code = ("import sys, os; sys.argv=['bench.py']; sys.path.insert(0, %r); import bench;"
            "import subprocess; real = subprocess.Popen;\n"
            "def fake(cmd, env=None, stdout=None):\n"
            "    r = int(env['RANK']); return real([sys.executable, '-c', 'import sys, time; time.sleep(0 if %%d else 30); sys.exit(%%d)' %% (r, 3 * r)])\n"
            "bench.subprocess.Popen = fake; rc = bench.launch_ranks(2, []); sys.exit(rc)") % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60, env=_clean_env())
    assert out.returncode == 3, (out.returncode, out.stderr[-500:])     # rank 1 exits 3 at once; rank 0 (sleeping) is terminated, not waited for


def test_closure_guesses_and_chi2_match_helpers():
    sys.path.insert(0, ROOT)
    import bench
    from g2o_frontend_amd import synth
    g = bench.closure_guesses([0, 5, 77])
    assert g.shape == (3, 16) and g.dtype == np.float32
    for i, s in enumerate([0, 5, 77]):
        T = g[i].reshape(4, 4).T
        assert T[2, 3] == 0.0 and np.array_equal(T[3], [0, 0, 0, 1])                 # pwn_matcher_base.cpp:114
        true = synth.pair_pose(s)
        assert np.abs(T[:2, 3] - true[:2, 3]).max() <= 0.012 and np.abs(T[:3, :3] - true[:3, :3]).max() < 0.02
        assert not np.allclose(T[:3, :3], np.eye(3), atol=1e-4)                      # non-identity: the batch shortcut cannot fire
    from g2o_frontend_amd.api import ALIGN_RESULT_DTYPE
    res = np.zeros(2, ALIGN_RESULT_DTYPE)
    res["iterations"] = 3; res["chi2"][0, :3] = (100.0, 50.0, 25.0); res["T"][0] = np.eye(4, dtype=np.float32).reshape(-1)
    tr = [{"seed": 0, "chi2_fp64": [100.0, 50.0005, 25.0], "chi2_fp32_serial": [100.0, 50.0, 25.1], "T": np.eye(4).tolist()}]
    m = bench.chi2_match(tr, res)                                                  # without a workload: free-running comparison only
    assert abs(m["free_running_max_rel_diff_vs_fp64_accumulated_oracle"] - 0.0005 / 50.0005) < 1e-9 and m["ok"] and m["free_running_max_abs_pose_diff"] == 0.0
    assert abs(m["free_running_max_rel_diff_vs_reference_fp32_serial_sums"] - 0.1 / 25.1) < 1e-6
    tr[0]["chi2_fp64"][1] = 50.1
    assert not bench.chi2_match(tr, res)["ok"]


def test_render_pool_keeps_job_order():
    sys.path.insert(0, ROOT)
    import bench
    from g2o_frontend_amd import synth
    K = synth.scaled_K(synth.K_VGA, 4)
    jobs = [("pair", s, 60, 80, K) for s in range(5)] + [("frame", 9, np.eye(4).tolist(), 60, 80, K, 3)]
    got = bench.render_all(jobs, world=1)
    for s in range(5):
        ref, cur, _ = synth.make_pair(s, 60, 80, K)
        assert np.array_equal(got[s][0], ref) and np.array_equal(got[s][1], cur)
    assert np.array_equal(got[5], synth.render_depth_mm(9, np.eye(4), 60, 80, K, hole_stream=3))


def test_strong_scaling_mode_shards_one_pair_list_and_assembles_the_same_records():
    """`--total-pairs T` (BASELINE configs[3] literally: the SAME 1024-pair list over N GPUs): the records rank 0 assembles at N = 2 --
    uneven shards included -- are the records of the N = 1 run (CRC per record), and the line says "strong"."""
    runs = {}
    for n in (1, 2):
        out = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--dry-run-cpu", "--total-pairs", "13"], capture_output=True, text=True,
                             timeout=300, env=_clean_env())
        assert out.returncode == 0, out.stderr[-2000:]
        runs[n] = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        assert runs[n]["records"] == 13 and runs[n]["records_ok"] and runs[n]["scaling"] == "strong" and runs[n]["total_pairs"] == 13
    assert runs[1]["pairs_per_gpu"] == 13 and runs[2]["pairs_per_gpu"] in (6, 7)
    assert runs[1]["records_crc"] == runs[2]["records_crc"]


def test_world_8_dry_runs_weak_and_strong_assemble_the_records_of_the_single_rank_run():
    """the driver's N = 8 launch rehearsed on the CPU (gloo): weak scaling = BASELINE configs[3] (8 x 128 pair ids) and strong scaling
    (--total-pairs 1024 over 8 ranks) both assemble, on rank 0, exactly the 1024 records of the one-rank run; every rank pinned to its
    own block of cores when the box has enough of them"""
    runs = {}
    for key, argv in (("n1", ["--gpus", "1", "--total-pairs", "1024"]), ("weak8", ["--gpus", "8", "--pairs", "128"]), ("strong8", ["--gpus", "8", "--total-pairs", "1024"]),
                      ("strong4", ["--gpus", "4", "--total-pairs", "1024"])):
        out = subprocess.run([sys.executable, BENCH, "--dry-run-cpu"] + argv, capture_output=True, text=True, timeout=600, env=_clean_env())
        assert out.returncode == 0, out.stderr[-2000:]
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, out.stdout
        runs[key] = json.loads(lines[0])
        assert runs[key]["records"] == 1024 and runs[key]["records_ok"] and runs[key]["total_pairs"] == 1024
    assert runs["weak8"]["n_gpus"] == 8 and runs["weak8"]["pairs_per_gpu"] == 128 and runs["weak8"]["scaling"] == "weak" and runs["weak8"]["max_rank_seen"] == 7
    assert runs["strong8"]["scaling"] == "strong" and runs["strong8"]["pairs_per_gpu"] == 128 and runs["strong4"]["pairs_per_gpu"] == 256
    assert runs["n1"]["records_crc_all"] == runs["weak8"]["records_crc_all"] == runs["strong8"]["records_crc_all"] == runs["strong4"]["records_crc_all"]
    ncpu = len(os.sched_getaffinity(0))
    if ncpu >= 8:
        assert runs["strong8"]["rank0_cpus"] == ncpu // 8 and runs["n1"]["rank0_cpus"] == ncpu


def test_the_drivers_launch_line_through_torch_distributed_run():
    """the N > 1 launch exactly as the driver issues it: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...: ranks read RANK / LOCAL_RANK / WORLD_SIZE / LOCAL_WORLD_SIZE from the env, pin themselves, rank 0 prints
    the one line"""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                          BENCH, "--gpus", "2", "--dry-run-cpu", "--pairs", "128"], capture_output=True, text=True, timeout=600, env=_clean_env())
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["records"] == 256 and j["records_ok"] and j["max_rank_seen"] == 1 and j["scaling"] == "weak"
    ncpu = len(os.sched_getaffinity(0))
    if ncpu >= 2:
        assert j["rank0_cpus"] == ncpu // 2


def test_rank_cpu_blocks_are_disjoint_and_follow_the_sockets():
    sys.path.insert(0, ROOT)
    import bench
    two = {0: list(range(0, 64)) + list(range(128, 192)), 1: list(range(64, 128)) + list(range(192, 256))}      # 2 sockets, hyper-threads numbered after the cores
    blocks = [bench.rank_cpus(r, 8, two) for r in range(8)]
    assert all(len(b) == 32 for b in blocks) and len(set(c for b in blocks for c in b)) == 256
    assert all(set(blocks[r]) <= set(two[0]) for r in range(4)) and all(set(blocks[r]) <= set(two[1]) for r in range(4, 8))
    one = {0: list(range(10))}
    b3 = [bench.rank_cpus(r, 3, one) for r in range(3)]
    assert sorted(c for b in b3 for c in b) == list(range(10)) and all(b for b in b3)
    assert bench.rank_cpus(0, 1, two) == [] and bench.rank_cpus(0, 16, {0: list(range(8))}) == []                   # one rank / fewer cpus than ranks: left alone
    odd = {0: list(range(6)), 1: list(range(6, 12))}
    assert [len(bench.rank_cpus(r, 3, odd)) for r in range(3)] == [4, 4, 4]                                          # ranks not divisible by sockets: flat split


def test_records_crc_gate(tmp_path, monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    rec = np.arange(5 * 20, dtype=np.float32).reshape(5, 20)
    monkeypatch.setattr(bench, "CRC_DIR", str(tmp_path))
    assert bench.check_records_crc(rec, 480, 640)["checked"] == 0                     # no file yet
    w = bench.check_records_crc(rec, 480, 640, write=True)
    assert w["checked"] == 5 and w["equal"] and w["file_is_for_these_kernels"]
    assert bench.check_records_crc(rec[:3], 480, 640)["checked"] == 3                 # a shorter pair list checks its prefix
    bad = rec.copy(); bad[2, 7] += 1
    r = bench.check_records_crc(bad, 480, 640)
    assert not r["equal"] and r["first_mismatch"] == 2 and r["mismatches"] == 1
    assert bench.check_records_crc(rec, 960, 1280)["checked"] == 0                    # another frame size: not comparable
    # one file per (workload, omega storage): the other three combinations do not see the one just written
    assert bench.check_records_crc(rec, 480, 640, omega_storage="sym6")["checked"] == 0
    assert bench.check_records_crc(rec, 480, 640, mode="partition")["checked"] == 0
    assert bench.check_records_crc(rec, 480, 640, omega_storage="sym6", mode="partition", write=True)["checked"] == 5
    assert sorted(os.listdir(tmp_path)) == ["records_crc_pairs_exact9.json", "records_crc_partition_sym6.json"]


def test_committed_records_crc_file_belongs_to_the_committed_kernels():
    """profiles/records_crc.json is only a gate for multi-GPU runs while its digest is the one of the sources in the tree: a change under
    g2o_frontend_amd/csrc/ without `bench.py --total-pairs 1024 --write-records-crc` on a GPU box leaves a stale file, which this test reports here
    instead of `file_is_for_these_kernels: false` in a scaling run."""
    import json
    import bench
    ref = json.load(open(bench.RECORDS_CRC_FILE))
    assert ref["kernel_source_digest"] == bench.kernel_source_digest(), "regenerate profiles/records_crc.json (see profiles/README.md)"
    assert ref["pairs"] == 1024 and len(ref["crc32"]) == 1024 and (ref["rows"], ref["cols"]) == (480, 640)


def test_partition_mode_byte_plumbing_world_2_and_8_gloo():
    """bench.py --mode partition (PwnCloser::processPartition sharded, `current` replicated by one broadcast) with the CPU dry run:
    launcher -> ranks -> a flat-cloud-sized byte pattern broadcast from rank 0 and checked on every rank -> 72-float match records (alignment
    record + the four score words) all-gathered and assembled by pair id.  World 2 and world 8 assemble the records of the one-rank run."""
    def run(n, pairs):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--dry-run-cpu", "--mode", "partition", "--total-pairs", str(pairs)],
                             capture_output=True, text=True, timeout=300, env=dict(os.environ, PWN_BENCH_NO_AFFINITY="1"))
        assert out.returncode == 0, out.stderr[-1500:]
        return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    one, two, three, eight = run(1, 24), run(2, 24), run(3, 24), run(8, 24)
    # flat_cloud_broadcast_ok also covers bench.simulate_partition_pipeline_cpu: the pipelined step's bookkeeping (ring of four flat buffers, owner rotation
    # j % world, sizes one step ahead in the owner's control row, import into the replica the next step matches against) with flat clouds of VARYING size,
    # eleven steps, every arrived keyframe compared byte for byte with what its owner made -- at world 3 (no divisor of the ring) and 8
    for r, n in ((one, 1), (two, 2), (three, 3), (eight, 8)):
        assert r["mode"] == "partition" and r["n_gpus"] == n and r["records"] == 24 and r["records_ok"] is True and r["flat_cloud_broadcast_ok"] is True
    assert one["records_crc_all"] == two["records_crc_all"] == three["records_crc_all"] == eight["records_crc_all"]


def test_pipeline_simulation_notices_a_wrong_protocol(monkeypatch):
    """the CPU simulation is not vacuous: with the size read from the wrong row it reports failure"""
    sys.path.insert(0, ROOT)
    import bench
    assert bench.simulate_partition_pipeline_cpu(0, 1, 9, None) is True
    monkeypatch.setattr(bench, "pp_ctrl_row", lambda owner, rows: 0)
    assert bench.simulate_partition_pipeline_cpu(0, 1, 9, None) is False


def test_host_cpu_info_names_what_cores_means():
    import bench
    info = bench.host_cpu_info()
    assert info["logical_cpus"] >= 1 and 1 <= info["physical_cores"] <= info["logical_cpus"] and isinstance(info["smt"], bool)
    assert "cgroup_cpu_quota" in info
