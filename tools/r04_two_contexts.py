#!/usr/bin/env python3
"""Experiment (round 4): does a second context driven by a second host thread hide the host gap between two steps (0.15-0.2 ms of the 10.3 ms
step: wake-up, result fill, descriptor preparation, descriptor uploads)?  One context with 128 pairs per step against two contexts with 64 pairs
each, every context stepping on its own thread (ctypes releases the GIL during the calls); same total work per unit of time."""
import os
import sys
import threading
import time
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402


def main():
    import torch
    torch.cuda.set_device(0)
    rows, cols, P = 480, 640, 128
    K, conv, alig = bench.conf(rows, cols)
    frames = bench.render_all([("pair", s, rows, cols, K) for s in range(P)], 1)
    args = types.SimpleNamespace(streams=2, sub_frames=64, sub_pairs=64, omega_storage="sym6", step_mode="fused")

    def run(workloads, seconds=3.0):
        counts = [0] * len(workloads)
        stop = time.perf_counter() + seconds

        def loop(i):
            while time.perf_counter() < stop:
                workloads[i].step(); counts[i] += 1
        for w in workloads:
            w.step(); w.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        th = [threading.Thread(target=loop, args=(i,)) for i in range(len(workloads))]
        for t in th: t.start()
        for t in th: t.join()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        return sum(c * w.P for c, w in zip(counts, workloads)) / dt

    one = bench.BatchWorkload(args, 0, rows, cols, P, frames, list(range(P)), False, 1)
    r1 = [run([one]) for _ in range(2)]
    one.close()
    halves = [bench.BatchWorkload(args, 0, rows, cols, P // 2, frames[i * P // 2:(i + 1) * P // 2], list(range(i * P // 2, (i + 1) * P // 2)), False, 1) for i in range(2)]
    r2 = [run(halves) for _ in range(2)]
    for w in halves: w.close()
    two_full = [bench.BatchWorkload(args, 0, rows, cols, P, frames, list(range(P)), False, 1) for _ in range(2)]
    r3 = [run(two_full) for _ in range(2)]
    print("one context x 128 pairs: %s alignments/s; two contexts x 64 pairs: %s; two contexts x 128 pairs: %s" % ([round(x) for x in r1], [round(x) for x in r2], [round(x) for x in r3]))


if __name__ == "__main__":
    main()
