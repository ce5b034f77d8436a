"""The multi-rank step rehearsed on the ONE GPU a test box has: two real rank processes (started by bench.py's own launcher before anything touches
the GPU), each with its own HIP context, streams and library instance on device 0, the collectives through gloo (RCCL refuses two ranks on one
device).  What it exercises that the world-size-1 tests cannot: two shards whose records interleave in the all-gather, the pwn_hip_ctx_wait_stream /
_signal_stream ordering around the record buffers with ANOTHER process's kernels contending for the device, rank 0's look-ahead job and the
broadcast of its flat cloud to a rank that really is another process (pwn_tracker/pwn_closer.cpp:85-111 sharded).
Gate: the assembled records equal, CRC by CRC, the digests committed for a single-GPU run (profiles/records_crc*.json)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(extra, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "PWN_BENCH_FORCE_DIST")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--one-device", "--pairs", "64", "--steps", "3", "--warmup", "1",
                          "--no-cpu-baseline", "--no-latency", "--no-extras"] + extra, env=env, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stderr[-3000:]
    return json.loads(out.stdout.strip().splitlines()[-1]), out.stderr


def _common(line, err, mode):
    assert line["n_gpus"] == 2 and line["config"]["mode"] == mode and "ONE device" in line["config"]["rehearsal"]
    assert line["config"]["pairs_per_gpu"] == 64 and line["config"]["total_pairs"] == 128
    g = line["gather"]
    assert g["backend"].startswith("gloo") and g["world"] == 2 and g["records"] == 128 and g["records_equal_local"] is True
    c = g["records_vs_single_gpu_run"]
    assert c["checked"] == 128 and c["equal"] is True and c["mismatches"] == 0 and c["file_is_for_these_kernels"] is True, c
    m = line["multi_gpu"]
    assert len(m["per_rank_ms_per_step"]) == 2 and all(x > 0 for x in m["per_rank_ms_per_step"])
    assert len(m["rank_cpus"]) == 2 and m["collectives_alone"]["gather_ms"] > 0
    assert line["value"] > 0 and 0.0 < line["roofline"]["frac"] < 1.0
    # both ranks reported their phases (bench.py's per-rank progress lines)
    assert "[bench rank 0/2" in err and "[bench rank 1/2" in err


def test_pairs_mode_two_ranks_on_one_device():
    line, err = _run([])
    _common(line, err, "pairs")
    assert line["scaling"] == "weak"


@pytest.mark.parametrize("serial", [False, True])
def test_partition_mode_two_ranks_on_one_device(serial):
    """rank 1 matches against a replica that arrived by broadcast from another process; pipelined (look-ahead job, broadcast and import queued from
    inside the match call) and the serial chain give the same records"""
    line, err = _run(["--mode", "partition"] + (["--partition-serial"] if serial else []))
    _common(line, err, "partition")
    p = line["partition"]
    assert p["pipelined"] is (not serial) and p["accepted_by_closer_thresholds_rank0"] >= 50
    assert 10e6 < p["flat_cloud_bytes"] <= p["flat_buffer_bound_bytes"]
    if not serial:
        assert p["broadcast_bytes_per_step"] == p["flat_cloud_bytes"]
        assert p["pipeline"]["rank0_lookahead_job_ms_per_step"] > 0
    assert line["multi_gpu"]["collectives_alone"]["broadcast_ms"] > 0


def test_partition_mode_three_ranks_rotate_the_look_ahead():
    """three ranks (not a divisor of the ring of four flat buffers): keyframe j is converted, exported and broadcast by rank j % 3, every rank imports
    every keyframe; the records still assemble to the committed digests"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "PWN_BENCH_FORCE_DIST")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--one-device", "--mode", "partition", "--pairs", "32", "--steps", "5", "--warmup", "2",
                          "--no-cpu-baseline", "--no-latency", "--no-extras"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    c = line["gather"]["records_vs_single_gpu_run"]
    assert line["n_gpus"] == 3 and line["gather"]["records"] == 96 and c["checked"] == 96 and c["equal"] is True and c["file_is_for_these_kernels"] is True, c
    p = line["partition"]
    assert p["pipelined"] is True and p["pipeline"]["lookahead_rotates_over_ranks"] is True and p["pipeline"]["rank0_lookahead_job_ms_per_step"] > 0
