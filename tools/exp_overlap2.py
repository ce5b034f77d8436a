#!/usr/bin/env python3
"""Experiment: do a converter call (VALU / store bound kernels) and an aligner call (HBM-read bound kernels) from two contexts / host
threads overlap on the device?  Each alone, then both at once; reports calls/s of each."""
import ctypes as C
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402


def main():
    import torch
    from g2o_frontend_amd import api, synth
    rows, cols = 480, 640
    N = rows * cols
    K, conv, alig = bench.conf(rows, cols)
    torch.cuda.set_device(0)
    P = 64
    frames = bench.render_all([("pair", s, rows, cols, K) for s in range(16)], 1)

    def make(streams):
        ctx = api.Context(device=0, max_rows=rows, max_cols=cols, max_batch=2 * 64)
        ctx.set_subbatch(64, 64); ctx.set_concurrency(streams); ctx.set_profiling(False)
        converter, aligner = bench.build_objects(ctx, rows, cols, K, conv, alig)
        rd = [torch.from_numpy(frames[i % 16][0].view(np.int16)).cuda() for i in range(P)]
        cd = [torch.from_numpy(frames[i % 16][1].view(np.int16)).cuda() for i in range(P)]
        refs = [api.Cloud(ctx, N) for _ in range(P)]; curs = [api.Cloud(ctx, N) for _ in range(P)]
        cprep = converter.batchHandles(refs + curs, rd + cd)
        aprep = ((C.c_void_p * P)(*[c.h for c in refs]), (C.c_void_p * P)(*[c.h for c in curs]), P)
        conv_call = lambda: converter.computeBatch(refs + curs, rd + cd, raw_scale=0.001, prepared=cprep)
        align_call = lambda: aligner.alignBatch(refs, curs, raw=True, prepared=aprep)
        conv_call(); align_call()
        return conv_call, align_call, (ctx, rd, cd, refs, curs)

    for streams in (1, 2):
        cA, aA, keepA = make(streams)
        cB, aB, keepB = make(streams)

        def loop(fn, secs, out, key):
            n, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < secs:
                fn(); n += 1
            out[key] = n / (time.perf_counter() - t0)

        res = {}
        loop(cA, 1.0, res, "convert_alone"); loop(aB, 1.0, res, "align_alone")
        ta = threading.Thread(target=loop, args=(cA, 2.0, res, "convert_with_align")); tb = threading.Thread(target=loop, args=(aB, 2.0, res, "align_with_convert"))
        ta.start(); tb.start(); ta.join(); tb.join()
        ta = threading.Thread(target=loop, args=(cA, 2.0, res, "convert_with_convert")); tb = threading.Thread(target=loop, args=(cB, 2.0, res, "convert_with_convert_b"))
        ta.start(); tb.start(); ta.join(); tb.join()
        ta = threading.Thread(target=loop, args=(aA, 2.0, res, "align_with_align")); tb = threading.Thread(target=loop, args=(aB, 2.0, res, "align_with_align_b"))
        ta.start(); tb.start(); ta.join(); tb.join()
        ms = {k: round(1e3 / v, 3) for k, v in res.items()}
        serial = ms["convert_alone"] + ms["align_alone"]
        # in T seconds of the concurrent run: conv calls = T/ms_c, align calls = T/ms_a; the same work serially takes T/ms_c*conv_alone + T/ms_a*align_alone
        gain = ms["convert_alone"] / ms["convert_with_align"] + ms["align_alone"] / ms["align_with_convert"]
        print(f"streams={streams} ms per call (128 frames / 64 pairs): {ms}  serial sum {serial:.2f}  concurrent speed-up x{gain:.3f}", flush=True)
        del keepA, keepB


if __name__ == "__main__":
    main()
