#!/usr/bin/env python3
"""Experiment: how much does overlapping the converter (VALU / store bound) with the aligner (HBM-read bound) gain?

Two contexts (own streams each), two host threads, each owning half of the pairs and running convert -> align per step.
Mode `serial`: one context does everything (the bench.py product configuration).  Mode `phase`: thread B starts half a step
late, so that one context converts while the other aligns.  Prints alignments/s for both.
"""
import argparse
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=128)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--sub", type=int, default=64)
    ap.add_argument("--delay-ms", type=float, default=3.0)
    a = ap.parse_args()
    import ctypes as C
    import torch
    from g2o_frontend_amd import api, synth
    rows, cols = 480, 640
    N = rows * cols
    K, conv, alig = bench.conf(rows, cols)
    torch.cuda.set_device(0)

    def make_worker(P, seeds, sub, streams):
        ctx = api.Context(device=0, max_rows=rows, max_cols=cols, max_batch=2 * sub)
        ctx.set_subbatch(sub, sub); ctx.set_concurrency(streams); ctx.set_profiling(False)
        converter, aligner = bench.build_objects(ctx, rows, cols, K, conv, alig)
        rd, cd = [], []
        for s in seeds:
            r, c, _ = synth.make_pair(s, rows, cols, K)
            rd.append(torch.from_numpy(r.view(np.int16)).cuda()); cd.append(torch.from_numpy(c.view(np.int16)).cuda())
        refs = [api.Cloud(ctx, N) for _ in range(P)]; curs = [api.Cloud(ctx, N) for _ in range(P)]
        cprep = converter.batchHandles(refs + curs, rd + cd)
        aprep = ((C.c_void_p * P)(*[c.h for c in refs]), (C.c_void_p * P)(*[c.h for c in curs]), P)
        out = {}

        def step():
            converter.computeBatch(refs + curs, rd + cd, raw_scale=0.001, prepared=cprep)
            out["res"] = aligner.alignBatch(refs, curs, raw=True, prepared=aprep)
        return step, out, (ctx, rd, cd, refs, curs)

    P = a.pairs
    # serial: one context, all pairs
    step, out, keep = make_worker(P, list(range(P)), a.sub, 2)
    step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"serial  one context, {P} pairs/step, sub {a.sub}, 2 streams: {P * a.steps / dt:8.0f} alignments/s  ({dt / a.steps * 1e3:.2f} ms/step)", flush=True)
    chi_serial = out["res"]["error"].copy()
    del step, out, keep

    # phase-shifted: two contexts with P/2 pairs each
    for sub, streams in ((a.sub // 2, 2), (a.sub, 1), (a.sub // 2, 1)):
        h = P // 2
        sA, oA, kA = make_worker(h, list(range(0, h)), sub, streams)
        sB, oB, kB = make_worker(h, list(range(h, P)), sub, streams)
        sA(); sB(); torch.cuda.synchronize()

        def loop(fn, delay):
            time.sleep(delay)
            for _ in range(a.steps):
                fn()
        for delay in (0.0, a.delay_ms * 1e-3):
            tA = threading.Thread(target=loop, args=(sA, 0.0)); tB = threading.Thread(target=loop, args=(sB, delay))
            t0 = time.perf_counter()
            tA.start(); tB.start(); tA.join(); tB.join()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0 - delay * 0      # the delay is part of the cost
            ok = np.array_equal(np.concatenate([oA["res"]["error"], oB["res"]["error"]]), chi_serial)
            print(f"overlap two contexts x {h} pairs, sub {sub}, {streams} stream(s) each, B delayed {delay * 1e3:.1f} ms: "
                  f"{P * a.steps / dt:8.0f} alignments/s  ({dt / a.steps * 1e3:.2f} ms/step)  same chi2: {ok}", flush=True)
        del sA, oA, kA, sB, oB, kB


if __name__ == "__main__":
    main()
