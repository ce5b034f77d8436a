"""The per-GPU shard shapes of BASELINE configs[3] and configs[4] through the two-call sequence -- one `computeBatch` + one `alignBatch` over
the whole shard, sub-batches of 64 dealt over two HIP streams, exact9 clouds -- as -m gpu tests (the configuration bench.py ships -- sym6 clouds,
one submission per step, four streams -- has its own oracle test: tests/test_gpu_step.py::test_the_shipped_configuration_against_the_oracle):

  * configs[3]: 128 VGA pairs per GPU (the 1024-pair loop-closure batch of pwn_tracker/pwn_closer.cpp:92-111 over 8 GPUs);
  * configs[4]: 32 pairs of 1280x960 per GPU (batch 256 over 8 GPUs).

Per shard: size-independent properties on every pair; bitwise equality with single alignments (Aligner::align on the same clouds, one
at a time) on sampled pairs; the converter's clouds bit-exact against the oracle and every iteration of the oracle's chi2 trace
re-run from the oracle's own iterate (teacher-forced: counters exact, chi2 1e-5) on sampled pairs; and the gather of the result
records through RCCL on device tensors with a process group of world size 1 (`PWN_BENCH_FORCE_DIST=1` in bench.py), so that the first
multi-GPU run is not also the first RCCL run."""
import concurrent.futures as cf
import os

import numpy as np
import pytest

from conftest import case_params

pytestmark = pytest.mark.gpu


def _render(name, seeds):
    """the shard's frames; numpy releases the GIL in the ray caster's array operations, so threads (no child process: this process
    has initialised the GPU) spread the rendering over the box's host cores"""
    from g2o_frontend_amd import synth
    rows, cols, K, _, _ = case_params(name)
    workers = max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 4))
    with cf.ThreadPoolExecutor(workers) as ex:
        return list(ex.map(lambda s: synth.make_pair(s, rows, cols, K), seeds))


def _bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype.itemsize == 4 else a


def _run_shard(name, seeds, singles, oracle_on, oracle, omega_storage="exact9"):
    from g2o_frontend_amd import api
    from test_gpu_parity import gpu_objects, oracle_params, _check_teacher_forced, _compare_clouds
    if omega_storage == "sym6":
        from test_omega_sym6 import compare_clouds_sym6
        _compare_clouds = lambda o, g, _name: compare_clouds_sym6(o, g)      # noqa: E731
    rows, cols, K, conv, alig = case_params(name)
    P, N = len(seeds), rows * cols
    pairs = _render(name, seeds)
    ctx = api.Context(0, rows, cols, 128, omega_storage=omega_storage)                     # 2 streams x 64 slots
    ctx.set_subbatch(64, 64); ctx.set_concurrency(2)
    _, converter, aligner = gpu_objects(ctx, name)
    frames = [ctx.upload(p[0]) for p in pairs] + [ctx.upload(p[1]) for p in pairs]      # resident uint16 frames, as in the timed region
    refs = [api.Cloud(ctx, N) for _ in range(P)]; curs = [api.Cloud(ctx, N) for _ in range(P)]
    converter.computeBatch(refs + curs, frames, raw_scale=0.001)
    res = aligner.alignBatch(refs, curs)
    assert len(res) == P
    # ---- properties on every pair of the shard
    worst_t = 0.0
    for i, (r, (_, _, Ttrue)) in enumerate(zip(res, pairs)):
        dt = float(np.abs(r["T"][:3, 3] - Ttrue[:3, 3]).max()); worst_t = max(worst_t, dt)
        assert dt < 5e-3 and np.abs(r["T"][:3, :3] - Ttrue[:3, :3]).max() < 5e-3, (i, dt)
        assert np.abs(r["T"][:3, :3] @ r["T"][:3, :3].T - np.eye(3)).max() < 1e-5 and np.array_equal(r["T"][3], [0, 0, 0, 1]), i
        assert r["iterations"] == 10 and r["chi2"][-1] < 0.2 * r["chi2"][0], i
        assert r["inliers"] > N // 3 and np.all(r["C"] <= r["K"]) and np.all(r["K"] <= N), i
        assert r["error"] == r["chi2"][-1] and r["inliers"] == r["iter_inliers"][-1], i
        assert refs[i].size() == int(((pairs[i][0] >= 500) & (pairs[i][0] <= 4500)).sum()), i      # point count = pixels inside [min, max] distance
    # ---- the whole shard again: run-to-run bitwise identical (two streams, rolling z-buffer tags)
    again = aligner.alignBatch(refs, curs)
    for a, b in zip(res, again):
        assert np.array_equal(_bits(a["T"]), _bits(b["T"])) and np.array_equal(_bits(a["chi2"]), _bits(b["chi2"]))
    # ---- sampled pairs one at a time: bitwise the batch results
    for i in singles:
        aligner.setReferenceCloud(refs[i]); aligner.setCurrentCloud(curs[i])
        g = aligner.align()
        for k in ("T", "chi2", "C", "K", "iter_inliers"):
            assert np.array_equal(_bits(g[k]), _bits(res[i][k])), (i, k)
    # ---- sampled pairs against the oracle: clouds bit-exact, every iteration teacher-forced
    cp, ap = oracle_params(oracle, name, accumulate_fp64=1)
    worst_chi2 = 0.0
    for i in oracle_on:
        ref = oracle.convert_16u_to_32f(pairs[i][0]); cur = oracle.convert_16u_to_32f(pairs[i][1])
        oref, _, _ = oracle.convert(cp, ref); ocur, _, _ = oracle.convert(cp, cur)
        _compare_clouds(oref.arrays(), refs[i].arrays(), name); _compare_clouds(ocur.arrays(), curs[i].arrays(), name)
        o = oracle.align(ap, oref, ocur)
        aligner.setReferenceCloud(refs[i]); aligner.setCurrentCloud(curs[i])
        worst_chi2 = max(worst_chi2, _check_teacher_forced(aligner, o))
        it0 = o["iterations"][0]
        assert (int(res[i]["K"][0]), int(res[i]["C"][0]), int(res[i]["iter_inliers"][0])) == (it0["K"], it0["C"], it0["inliers"]), i
        assert np.abs(res[i]["T"] - o["T"]).max() <= 1e-5, (i, np.abs(res[i]["T"] - o["T"]).max())
    print(f"{name} shard of {P} pairs ({omega_storage}): worst |t - t_true| {worst_t:.1e} m; {len(singles)} singles bitwise equal; "
          f"{len(oracle_on)} pairs vs oracle: clouds bit-exact, worst teacher-forced chi2 rel diff {worst_chi2:.1e}")
    for f in frames:
        f.free()
    ctx.close()
    return res


def test_config3_shard_128_vga_pairs(oracle):
    P = 128
    seeds = list(range(3000, 3000 + P))
    _run_shard("vga", seeds, singles=(0, 17, 40, 63, 64, 65, 100, 127), oracle_on=(5, 63, 64, 127), oracle=oracle)


def test_config4_shard_32_pairs_1280x960(oracle):
    P = 32
    seeds = list(range(5000, 5000 + P))
    _run_shard("k2", seeds, singles=(0, 15, 16, 31), oracle_on=(31,), oracle=oracle)


def test_result_gather_through_rccl_world_size_1():
    """bench.py's step with the gather forced through torch.distributed (backend nccl = RCCL, world size 1, device tensors): the
    records rank 0 assembles are the records of the shard, bit for bit.  A child process: the process group must be created before
    the HIP context of this test process exists in the child, and MASTER_* are the child's own."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, PWN_BENCH_FORCE_DIST="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--pairs", "6", "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline", "--no-latency", "--no-extras", "--render-workers", "1", "--check-gather"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["gather"]["backend"].startswith("nccl") and line["gather"]["forced"] is True
    assert line["gather"]["records_equal_local"] is True and line["gather"]["records"] == 6
    assert line["value"] > 0


def test_strong_mode_at_configs3_size_on_one_gpu_matches_the_committed_record_digests():
    """BASELINE configs[3] literally -- the 1024-pair list -- on ONE GPU in strong-scaling mode (`bench.py --gpus 1 --total-pairs 1024`): the
    records equal, CRC by CRC, those committed in profiles/records_crc.json (written by the same command with --write-records-crc for the
    kernel sources in the tree).  This is the gate a multi-GPU run is held to (`gather.records_vs_single_gpu_run`): a pair's record is the
    same bits whatever rank, sub-batch or position aligned it."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--total-pairs", "1024", "--steps", "1", "--warmup", "0",
                          "--no-cpu-baseline", "--no-latency", "--no-extras", "--no-profile"], env=env, capture_output=True, text=True, timeout=1100)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    g = line["gather"]["records_vs_single_gpu_run"]
    assert line["scaling"] == "strong" and line["config"]["total_pairs"] == 1024 and line["config"]["pairs_per_gpu"] == 1024
    assert g["checked"] == 1024 and g["file_is_for_these_kernels"] is True, g
    assert g["equal"] is True and g["mismatches"] == 0, g
    assert line["gather"]["records_equal_local"] is True
