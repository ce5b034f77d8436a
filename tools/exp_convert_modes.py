"""Converter wall time per 256 resident VGA frames (4 sub-batches of 64 over two streams) and the whole bench step under the converter's
A/B switch PWN_FUSED_CONVERT (k_convert_fused vs k_unproject_integral + k_stats), read when the context is created.  python tools/exp_convert_modes.py [reps] [mode ...]   with mode = VAR=VALUE[,VAR=VALUE]"""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from g2o_frontend_amd import api, synth

rows, cols = 480, 640
K, conv, alig = bench.conf(rows, cols)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
MODES = sys.argv[2:] or ["PWN_FUSED_CONVERT=0", "PWN_FUSED_CONVERT=1"]
SUBF = int(os.environ.get("PWN_SUB_FRAMES", 64))
P = 128
base = [synth.make_pair(s, rows, cols, K) for s in range(8)]
digest = {}
for rnd in range(2):
    for mode in MODES:
        for kv in mode.split(","):
            k, v = kv.split("="); os.environ[k] = v
        ctx = api.Context(0, rows, cols, 128); ctx.set_subbatch(SUBF, 64); ctx.set_concurrency(int(os.environ.get("PWN_STREAMS", 2)))
        converter, aligner = bench.build_objects(ctx, rows, cols, K, conv, alig)
        frames = [ctx.upload(base[i % 8][0]) for i in range(P)] + [ctx.upload(base[i % 8][1]) for i in range(P)]
        refs = [api.Cloud(ctx, rows * cols) for _ in range(P)]; curs = [api.Cloud(ctx, rows * cols) for _ in range(P)]
        prep = converter.batchHandles(refs + curs, frames)
        for _ in range(3):
            converter.computeBatch(refs + curs, None, raw_scale=0.001, prepared=prep)
        ctx.synchronize(); t = time.perf_counter()
        for _ in range(reps):
            converter.computeBatch(refs + curs, None, raw_scale=0.001, prepared=prep)
        ctx.synchronize(); conv_ms = (time.perf_counter() - t) / reps * 1e3
        res = aligner.alignBatch(refs, curs, raw=True)
        ctx.synchronize(); t = time.perf_counter()
        for _ in range(reps):
            converter.computeBatch(refs + curs, None, raw_scale=0.001, prepared=prep)
            res = aligner.alignBatch(refs, curs, raw=True)
        ctx.synchronize(); step_ms = (time.perf_counter() - t) / reps * 1e3
        a = refs[5].arrays()
        import hashlib
        h = hashlib.sha256(b"".join(np.ascontiguousarray(a[k]).tobytes() for k in sorted(a)) + res["T"].tobytes() + res["chi2"].tobytes()).hexdigest()[:12]
        digest.setdefault(h, []).append(mode)
        print(json.dumps({"mode": mode, "convert_ms_per_256_frames": round(conv_ms, 3), "step_ms": round(step_ms, 3), "alignments_per_s": round(P / step_ms * 1e3), "digest": h}), flush=True)
        for f in frames:
            f.free()
        ctx.close()
print("results identical across modes:", len(digest) == 1)
