"""Converter-only A/B on the GPU box: for every library in build/variants, `rounds` child processes in turn, each timing the converter's
kernel stages over `reps` 256-frame batches (serial profiled launches).  python tools/ab_convert.py [rounds] [reps]"""
import glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys, time, json
import numpy as np
sys.path.insert(0, %r)
import bench
from g2o_frontend_amd import api, synth
rows, cols = 480, 640
K, conv, alig = bench.conf(rows, cols)
n, reps = 256, int(sys.argv[1])
ctx = api.Context(0, rows, cols, 128); ctx.set_subbatch(64, 64)
converter, aligner = bench.build_objects(ctx, rows, cols, K, conv, alig)
base = [synth.make_pair(s, rows, cols, K)[0] for s in range(4)]
res = [ctx.upload(base[i %% 4]) for i in range(n)]
clouds = [api.Cloud(ctx, rows * cols) for _ in range(n)]
prep = converter.batchHandles(clouds, res)
ctx.set_concurrency(1); ctx.set_profiling(True)
acc = {}
for it in range(reps + 2):
    converter.computeBatch(clouds, None, raw_scale=0.001, prepared=prep)
    if it >= 2:
        for k in ("unproject", "integral", "stats"):
            acc[k] = acc.get(k, 0.0) + ctx.stage_ms(k)[0]
print(json.dumps({k: v / reps for k, v in acc.items()}))
''' % ROOT

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
reps = sys.argv[2] if len(sys.argv) > 2 else "10"
libs = sorted(glob.glob(os.path.join(ROOT, "build", "variants", "*.so")))
tot = {l: [] for l in libs}
for r in range(rounds):
    for l in libs:
        out = subprocess.run([sys.executable, "-c", CHILD, reps], env=dict(os.environ, PWN_HIP_LIB=l), capture_output=True, text=True, timeout=300)
        if out.returncode != 0:
            print(l, "FAILED", out.stderr[-300:]); continue
        d = json.loads(out.stdout.strip().split("\n")[-1]); tot[l].append(d)
        print(os.path.basename(l), {k: round(v, 3) for k, v in d.items()}, flush=True)
for l in libs:
    if tot[l]:
        print("mean", os.path.basename(l), {k: round(sum(d[k] for d in tot[l]) / len(tot[l]), 3) for k in tot[l][0]})
