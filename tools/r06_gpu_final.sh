#!/bin/bash
# round 6, final tree: full GPU suite as the driver runs it, smoke, the default line, the partition lines (plain, RCCL at world size 1, the serial chain)
set -u
O=gpurun_out/r06final; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== full GPU suite"; date
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/gpu_suite.txt 2>&1; echo "rc $?"; tail -4 $O/gpu_suite.txt
echo "== smoke"; timeout -k 10 200 python __graft_entry__.py --smoke 2>&1 | tail -2
echo "== bench lines"; date
timeout -k 10 500 python bench.py --steps 20 --warmup 3 > $O/bench_default.json 2> $O/bench_default.err; echo "rc $?"
timeout -k 10 300 python bench.py --mode partition --steps 30 --warmup 3 --no-cpu-baseline > $O/bench_partition.json 2> $O/bench_partition.err; echo "rc $?"
PWN_BENCH_FORCE_DIST=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29541 timeout -k 10 300 python bench.py --mode partition --steps 30 --warmup 3 --no-cpu-baseline > $O/bench_partition_rccl.json 2> $O/bench_partition_rccl.err; echo "rc $?"
PWN_BENCH_FORCE_DIST=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29542 timeout -k 10 300 python bench.py --mode partition --partition-serial --steps 30 --warmup 3 --no-cpu-baseline > $O/bench_partition_rccl_serial.json 2> $O/bench_partition_rccl_serial.err; echo "rc $?"
PWN_BENCH_FORCE_DIST=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29543 timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_default_rccl.json 2> $O/bench_default_rccl.err; echo "rc $?"
python - <<'PY'
import json
for f in ("bench_default", "bench_default_rccl", "bench_partition", "bench_partition_rccl", "bench_partition_rccl_serial"):
    try:
        l = json.loads(open(f"gpurun_out/r06final/{f}.json").read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "ERR", e); continue
    print(f, "%.0f/s %.3f ms kernel %.3f path %.3f" % (l["value"], l["ms_per_step"], l["roofline"]["frac"], l["roofline"]["path_frac"]), l["gather"]["records_vs_single_gpu_run"]["equal"], l["gather"]["records_vs_single_gpu_run"]["file_is_for_these_kernels"])
    for k in ("align_only", "closure_match_batch", "omega_exact9"):
        if k in l: print("   ", k, round(l[k]["alignments_per_s"]), l[k].get("frac_of_peak", l[k].get("path_frac")))
    if "config5_1280x960" in l:
        c = l["config5_1280x960"]; print("    config5", round(c["alignments_per_s"]), c["roofline"]["path_frac"])
    if "cpu_baseline" in l and l["cpu_baseline"]: print("    cpu", {k: l["cpu_baseline"].get(k) for k in ("value", "cores", "single_thread_value", "logical_cpus", "physical_cores", "cgroup_cpu_quota")})
    if "tracker_config2" in l: print("    tracker", round(l["tracker_config2"]["frames_per_s"]), l["tracker_config2"].get("roofline_frac"), l.get("single_pair_latency_ms"), l.get("single_pair_roofline_frac"), l.get("single_pair_latency_ms_cpp_mirror"))
    if "partition" in l: print("    partition", json.dumps(l["partition"].get("pipeline")), l["roofline"].get("traffic_over_algorithmic"), l["partition"].get("broadcast_bytes_per_step"))
    if l.get("multi_gpu"): print("    collectives", l["multi_gpu"]["collectives_alone"])
PY
date
