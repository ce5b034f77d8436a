#!/usr/bin/env python3
"""Headline benchmark: depth-pair alignments/second (640x480, 10 Gauss-Newton iterations) on N MI355X.

Workload (BASELINE.json configs[3] sharded as SURVEY.md §8(e) prescribes; at N=8 it is exactly the
1024-pair loop-closure batch, at N=1 it is one GPU's 128-pair shard): every rank owns `--pairs`
independent synthetic VGA depth pairs, resident in HBM as uint16 millimetre frames when the timed region
starts.  One step = for every pair: DepthImage_convert_16UC1_to_32FC1 + DepthImageConverterIntegralImage::compute
on both frames, then Aligner::align (10 outer x 1 inner iterations), then (N>1) an RCCL all-gather of the
4x4 poses.  Nothing is cached between steps.

Prints ONE JSON line (rank 0) with the contract fields plus `roofline` (dominant kernel, live hipEvent
timing inside the timed region) and `cpu_baseline` (the CPU oracle = a port of the reference path, timed on
this box's host cores on a bounded sample of the same workload).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pairs", type=int, default=128, help="depth pairs per GPU (weak scaling)")
    ap.add_argument("--rows", type=int, default=480)
    ap.add_argument("--cols", type=int, default=640)
    ap.add_argument("--sub-frames", type=int, default=int(os.environ.get("PWN_SUB_FRAMES", 64)))
    ap.add_argument("--sub-pairs", type=int, default=int(os.environ.get("PWN_SUB_PAIRS", 64)))
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency", action="store_true")
    ap.add_argument("--align-only", action="store_true", help="also time align-only (clouds resident)")
    ap.add_argument("--cpu-baseline-only", action="store_true", help="(internal) run only the CPU baseline leg and print its JSON")
    ap.add_argument("--no-profile", action="store_true", help="skip the serial profiled pass (roofline fields become 0)")
    ap.add_argument("--streams", type=int, default=2, help="HIP streams the batch calls use in the timed region (1 = serial)")
    return ap.parse_args()


def conf(rows, cols):
    from g2o_frontend_amd import synth
    from oracle import oracle as O   # parameter tables only (the oracle is the checker / cpu_baseline, never the product path)
    if (rows, cols) == (960, 1280):
        K = synth.K_1280
        conv = dict(O.VGA_CONF_CONVERTER, min_image_radius=20, max_image_radius=60, min_points=200)
    else:
        K = synth.K_VGA if (rows, cols) == (480, 640) else synth.scaled_K(synth.K_VGA, 640 // cols)
        conv = dict(O.VGA_CONF_CONVERTER)
    return K, conv, dict(O.VGA_CONF_ALIGNER)


def build_objects(ctx, rows, cols, K, conv, alig):
    from g2o_frontend_amd import api
    proj = api.PinholePointProjector()
    proj.setCameraMatrix([[K[0], 0, K[2]], [0, K[1], K[3]], [0, 0, 1]])
    proj.setMinDistance(conv["min_distance"]); proj.setMaxDistance(conv["max_distance"]); proj.setImageSize(rows, cols)
    st = api.StatsCalculatorIntegralImage()
    st.setWorldRadius(conv["world_radius"]); st.setMinImageRadius(conv["min_image_radius"]); st.setMaxImageRadius(conv["max_image_radius"])
    st.setMinPoints(conv["min_points"]); st.setCurvatureThreshold(conv["stats_curvature_threshold"])
    pi, ni = api.PointInformationMatrixCalculator(), api.NormalInformationMatrixCalculator()
    pi.setCurvatureThreshold(conv["point_info_curvature_threshold"]); ni.setCurvatureThreshold(conv["normal_info_curvature_threshold"])
    converter = api.DepthImageConverterIntegralImage(proj, st, pi, ni)
    f = api.CorrespondenceFinder()
    f.setInlierDistanceThreshold(alig["inlier_distance_threshold"]); f.setInlierNormalAngularThreshold(alig["inlier_normal_angular_threshold"])
    f.setFlatCurvatureThreshold(alig["flat_curvature_threshold"]); f.setInlierCurvatureRatioThreshold(alig["inlier_curvature_ratio_threshold"])
    f.setImageSize(rows, cols)
    lin = api.Linearizer(); lin.setInlierMaxChi2(alig["inlier_max_chi2"]); lin.setRobustKernel(alig["robust_kernel"])
    al = api.Aligner(ctx)
    al.setProjector(proj); al.setLinearizer(lin); al.setCorrespondenceFinder(f)
    al.setOuterIterations(alig["outer_iterations"]); al.setInnerIterations(alig["inner_iterations"])
    return converter, al


def cpu_baseline(rows, cols, K, conv, alig, seeds, budget_s):
    """The CPU oracle (port of the reference CPU path) on a bounded sample of the same workload, one thread."""
    from g2o_frontend_amd import synth
    from oracle import oracle as O
    O.set_num_threads(1)
    cp = O.converter_params(K=K, **conv)
    apar = O.aligner_params(rows, cols, K=K, **alig)
    done, t_conv, t_align, t0 = 0, 0.0, 0.0, time.perf_counter()
    for s in seeds:
        ref_mm, cur_mm, _ = synth.make_pair(s, rows, cols, K)
        a = time.perf_counter()
        ref = O.convert_16u_to_32f(ref_mm); cur = O.convert_16u_to_32f(cur_mm)
        cr, _, _ = O.convert(cp, ref); cc, _, _ = O.convert(cp, cur)
        b = time.perf_counter()
        O.align(apar, cr, cc)
        c = time.perf_counter()
        t_conv += b - a; t_align += c - b; done += 1
        if time.perf_counter() - t0 > budget_s:
            break
    tot = t_conv + t_align
    out = {"value": done / tot, "unit": "alignments/s", "cores": 1, "kind": "port",
           "sample": f"{done} of the benchmark's {rows}x{cols} pairs (convert 2 frames + align, 10 GN iterations), "
                     f"single thread, {tot:.1f} s CPU ({t_conv / done * 1e3:.0f} ms convert + {t_align / done * 1e3:.0f} ms align per pair)",
           "align_only_value": done / t_align}
    return out


def _cpu_worker(job):
    """one host core: `n` pairs of the same workload through the oracle, single-threaded; returns (pairs, seconds of oracle work)"""
    rows, cols, K, conv, alig, seeds = job
    from g2o_frontend_amd import synth
    from oracle import oracle as O
    O.set_num_threads(1)
    cp = O.converter_params(K=K, **conv)
    apar = O.aligner_params(rows, cols, K=K, **alig)
    frames = [synth.make_pair(s, rows, cols, K) for s in seeds]        # rendering the synthetic frames is not part of the measured work
    t0 = time.perf_counter()
    for ref_mm, cur_mm, _ in frames:
        cr, _, _ = O.convert(cp, O.convert_16u_to_32f(ref_mm)); cc, _, _ = O.convert(cp, O.convert_16u_to_32f(cur_mm))
        O.align(apar, cr, cc)
    return len(frames), time.perf_counter() - t0


def cpu_baseline_all_cores(rows, cols, K, conv, alig, per_core, max_cores=32):
    """the same oracle on every host core at once: independent pairs, one single-threaded process per core (the way the reference's own
    loop-closure batch would be spread over a CPU)"""
    import concurrent.futures as cf
    import multiprocessing as mp
    cores = max(1, min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), max_cores))
    jobs = [(rows, cols, K, conv, alig, [1000 + c * per_core + i for i in range(per_core)]) for c in range(cores)]
    with cf.ProcessPoolExecutor(max_workers=cores, mp_context=mp.get_context("spawn")) as ex:
        res = list(ex.map(_cpu_worker, jobs))
    pairs = sum(r[0] for r in res); slowest = max(r[1] for r in res)
    return {"value": pairs / slowest, "unit": "alignments/s", "cores": cores,
            "sample": f"{pairs} pairs, {per_core} per core on {cores} cores at once (one single-threaded oracle process per core), slowest core {slowest:.1f} s"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", 0)); world = int(os.environ.get("WORLD_SIZE", 1)); local = int(os.environ.get("LOCAL_RANK", 0))
    rows, cols, P = args.rows, args.cols, args.pairs
    N = rows * cols
    K, conv, alig = conf(rows, cols)
    n_it = alig["outer_iterations"] * alig["inner_iterations"]

    if args.cpu_baseline_only:
        out = cpu_baseline(rows, cols, K, conv, alig, list(range(0, 64)), args.cpu_seconds)
        try:      # all host cores beside the single-thread figure (bounded: as many pairs per core as ~cpu_seconds/2 of one core's work)
            per_core = max(1, int(0.5 * args.cpu_seconds * out["value"]))
            out["all_cores"] = cpu_baseline_all_cores(rows, cols, K, conv, alig, per_core)
        except Exception as e:      # never let the baseline leg break the benchmark line
            out["all_cores"] = {"error": str(e)[:200]}
        print(json.dumps(out))
        return
    # CPU baseline first (rank 0 only), in a child process started before anything touches the GPU: the process that
    # drives the GPU never loads the oracle library
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:      # at N = 1 only: the other ranks of a multi-GPU run would wait for it
        import subprocess
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--rows", str(rows), "--cols", str(cols),
                              "--cpu-seconds", str(args.cpu_seconds)], capture_output=True, text=True, timeout=600)
        try:
            cpu = json.loads(out.stdout.strip().splitlines()[-1])
        except Exception:
            cpu = {"error": (out.stderr or out.stdout)[-300:]}

    import torch
    import torch.distributed as dist
    from g2o_frontend_amd import api, shard, synth
    torch.cuda.set_device(local)
    use_dist = world > 1 or os.environ.get("PWN_BENCH_FORCE_DIST") == "1"      # FORCE_DIST: exercise the RCCL path at world size 1
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # RCCL prints a version banner on stdout when the communicator is created; keep stdout for the one JSON line
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

    slots = max(2, args.streams) * max(args.sub_frames, args.sub_pairs, 1)      # room for one sub-batch in flight per stream
    ctx = api.Context(device=local, max_rows=rows, max_cols=cols, max_batch=slots)
    ctx.set_subbatch(args.sub_frames, args.sub_pairs)
    converter, aligner = build_objects(ctx, rows, cols, K, conv, alig)

    # synthetic inputs of this rank's shard, uploaded to HBM (uint16 mm frames)
    my_pairs = shard.shard_range(world * P, rank, world)          # contiguous shard of the global pair list
    seeds = list(my_pairs)
    ref_dev, cur_dev = [], []
    for s in seeds:
        ref_mm, cur_mm, _ = synth.make_pair(s, rows, cols, K)
        ref_dev.append(torch.from_numpy(ref_mm.view(np.int16)).cuda())
        cur_dev.append(torch.from_numpy(cur_mm.view(np.int16)).cuda())
    refs = [api.Cloud(ctx, N) for _ in range(P)]
    curs = [api.Cloud(ctx, N) for _ in range(P)]
    records = torch.empty((P, shard.RECORD_FLOATS), dtype=torch.float32, device="cuda")

    stage_names = ["u16_to_f32", "unproject", "integral", "integral_rows", "integral_cols", "stats", "project", "corr_linearize", "solve"]
    stage_ms = {k: 0.0 for k in stage_names}
    stage_n = {k: 0 for k in stage_names}
    last = {}

    dbg = os.environ.get("PWN_BENCH_DEBUG")
    tm = {"convert": 0.0, "align": 0.0, "gather": 0.0}

    conv_prep = converter.batchHandles(refs + curs, ref_dev + cur_dev)       # handle / pointer arrays built once
    import ctypes as C
    align_prep = ((C.c_void_p * P)(*[c.h for c in refs]), (C.c_void_p * P)(*[c.h for c in curs]), P)
    records_host = torch.empty((P, shard.RECORD_FLOATS), dtype=torch.float32).pin_memory()

    def step(profile):
        t_a = time.perf_counter()
        converter.computeBatch(refs + curs, ref_dev + cur_dev, raw_scale=0.001, prepared=conv_prep)
        t_b = time.perf_counter()
        if profile:
            for k in stage_names[:6]:
                ms, n = ctx.stage_ms(k); stage_ms[k] += ms; stage_n[k] += n
        res = aligner.alignBatch(refs, curs, raw=True, prepared=align_prep)
        t_c = time.perf_counter()
        if profile:
            for k in stage_names[6:]:
                ms, n = ctx.stage_ms(k); stage_ms[k] += ms; stage_n[k] += n
        records_host.numpy()[:] = shard.pack_results_raw(res, seeds)
        records.copy_(records_host, non_blocking=False)
        last["gathered"] = shard.gather_records(records, world, P, force=use_dist)      # RCCL all-gather: the only collective of the path
        last["res"] = res
        if dbg and not profile:
            tm["convert"] += t_b - t_a; tm["align"] += t_c - t_b; tm["gather"] += time.perf_counter() - t_c

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # timed region: the product configuration (two streams, no event instrumentation)
    ctx.set_concurrency(args.streams)
    ctx.set_profiling(False)
    for _ in range(args.warmup):
        step(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(False)
    barrier()
    dt = time.perf_counter() - t0
    # per-kernel durations: the same K steps again on ONE stream with a hipEvent pair around every kernel stage (on the
    # library's stream).  With two streams, launches of different sub-batches overlap and a launch's elapsed time is no
    # longer the kernel's own duration, so the roofline figures come from this serial pass (rocprofv3 summaries in profiles/
    # are taken the same way: bench.py --streams 1).
    dt_serial = None
    if not args.no_profile:
        ctx.set_concurrency(1)
        ctx.set_profiling(True)
        step(False)
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step(True)
        barrier()
        dt_serial = time.perf_counter() - t1
        ctx.set_profiling(False)
        ctx.set_concurrency(args.streams)
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if dbg and rank == 0:
        print("host wall per step (ms, timed region + warmup):", {k: round(v / (args.steps + args.warmup) * 1e3, 2) for k, v in tm.items()}, file=sys.stderr)
    res = last["res"]
    if rank == 0:
        allrec = shard.assemble(last["gathered"].cpu().numpy(), world * P)      # every pair of every rank arrived exactly once
        assert allrec.shape[0] == world * P
    # measured counters of SURVEY.md §8(d): M_r, M_c, K_i, C_i -> algorithmic bytes
    Mr = res["n_reference"].astype(np.float64); Mc = res["n_current"].astype(np.float64)
    Ks = res["iter_candidates"].sum(1).astype(np.float64); Cs = res["iter_correspondences"].sum(1).astype(np.float64)
    bytes_convert = 2 * 8.0 * N + 64.0 * (Mr + Mc)                                   # two frames: 8N + 64M each
    bytes_fused = n_it * 8.0 * N + 72.0 * Ks + 28.0 * Cs                              # per pair, all iterations
    bytes_project = 16.0 * Mc + 4.0 * N + n_it * (16.0 * Mr + 4.0 * N)
    bytes_align = bytes_project + bytes_fused + 8.0 * N
    total_bytes_step = float((bytes_convert + bytes_align).sum())

    extra = {}
    if args.align_only:
        barrier(); a = time.perf_counter()
        for _ in range(args.steps):
            aligner.alignBatch(refs, curs, raw=True, prepared=align_prep)
        barrier(); extra["align_only_alignments_per_s"] = world * P * args.steps / (time.perf_counter() - a)
    if not args.no_latency and rank == 0:
        ctx.set_profiling(False)
        lat = []
        for _ in range(5):
            torch.cuda.synchronize(); a = time.perf_counter()
            converter.computeBatch([refs[0], curs[0]], [ref_dev[0], cur_dev[0]], raw_scale=0.001)
            aligner.alignBatch([refs[0]], [curs[0]])
            lat.append((time.perf_counter() - a) * 1e3)
        extra["single_pair_latency_ms"] = float(np.median(lat))

    hbm_read = hbm_copy = None
    if rank == 0:
        # the bandwidth this box actually delivers, next to the 8 TB/s spec figure (float4 streaming read / copy of 2 GiB)
        try:
            hbm_read, hbm_copy = ctx.measure_hbm(1 << 31)
        except Exception:
            hbm_read = hbm_copy = None
    if rank == 0:
        value = world * P * args.steps / dt
        launches = max(stage_n["corr_linearize"], 1)
        dom = max(stage_names, key=lambda k: stage_ms[k])
        # dominant kernel: the fused correspondence+linearize pass (one launch = one iteration of one sub-batch)
        k_ms = stage_ms["corr_linearize"] / launches
        k_bytes = float(bytes_fused.sum()) * args.steps / launches
        achieved = k_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tfile):
            try:
                # measured per pair-iteration by the PMC passes (profiles/), scaled to the pairs one launch of this run covers
                tj = json.load(open(tfile))
                # measured at VGA; every term of the kernel's traffic is per pixel / per point, so other frame sizes scale with the pixel count
                traffic = tj.get("k_corr_linearize_bytes_per_pair_iteration") * (N / tj.get("pixels_per_frame", 307200)) * (P * n_it * args.steps / launches)
            except Exception:
                traffic = None
        out = {
            "metric": "depth-pair alignments/sec (640x480, 10 GN iters)", "value": value, "unit": "alignments/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"loop-closure batch: {P} independent {cols}x{rows} depth pairs per GPU "
                                   f"(u16 mm frames resident in HBM; per pair: convert 2 frames + Aligner::align, "
                                   f"{alig['outer_iterations']}x{alig['inner_iterations']} GN iterations); BASELINE configs[3] shard",
                       "pairs_per_gpu": P, "rows": rows, "cols": cols, "sub_frames": args.sub_frames, "sub_pairs": args.sub_pairs,
                       "streams": args.streams,
                       "parallelism": f"independent pairs sharded over {world} GPU(s), RCCL all-gather of poses only"},
            "roofline": {"bound": "hbm", "kernel": "k_corr_linearize", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "bytes_per_launch_algorithmic": k_bytes, "avg_launch_ms": k_ms, "launches": stage_n["corr_linearize"],
                         "dominant_by_time": dom,
                         "measured_hbm_GBps": {"read": hbm_read, "copy": hbm_copy},    # this box, float4 streaming kernels over 2 GiB; not frac's denominator
                         "traffic_GBps": (traffic / (k_ms * 1e-3) / 1e9) if (traffic and k_ms > 0) else None,
                         "measured_in": "serial profiled pass of the same K steps (one stream, hipEvent pair per kernel stage); "
                                        "the timed region runs two streams without instrumentation",
                         "serial_pass_alignments_per_s": (world * P * args.steps / dt_serial) if dt_serial else None},
            "cpu_baseline": cpu,
            "path_roofline": {"algorithmic_bytes_per_pair": total_bytes_step / P, "achieved_GBps": total_bytes_step * args.steps / dt / 1e9,
                              "frac_of_peak": total_bytes_step * args.steps / dt / 1e9 / HBM_PEAK_GBS},
            "stage_ms_per_step": {k: stage_ms[k] / args.steps for k in stage_names},
            "stage_launches_per_step": {k: stage_n[k] / args.steps for k in stage_names},
            "counters_mean": {"M_ref": float(Mr.mean()), "M_cur": float(Mc.mean()), "K_sum": float(Ks.mean()), "C_sum": float(Cs.mean()),
                              "chi2_final": float(res["error"].mean()), "inliers_final": float(res["inliers"].mean())},
        }
        out.update(extra)
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
