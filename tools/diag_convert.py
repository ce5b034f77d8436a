#!/usr/bin/env python3
"""GPU-vs-oracle diagnostic: bitwise differences of the converter outputs and chi2 traces for one case."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import case_params, make_depth_pair
from test_gpu_parity import gpu_objects, oracle_params, upload
from g2o_frontend_amd import api
from oracle import oracle as O

name, seed = sys.argv[1], int(sys.argv[2])
rows, cols, K, conv, _ = case_params(name)
ref, cur, Ttrue, _, _ = make_depth_pair(name, seed)
ctx = api.Context(0, rows, cols, 2)
cp, ap = oracle_params(O, name, accumulate_fp64=1)
_, converter, aligner = gpu_objects(ctx, name)
oc = {}; gc = {}
for tag, d in (("ref", ref), ("cur", cur)):
    oc[tag], _, _ = O.convert(cp, d)
    gc[tag] = api.Cloud(ctx, rows * cols)
    converter.compute(gc[tag], d, keep_stats=True)
    o, g = oc[tag].arrays(stats=True), gc[tag].arrays(stats=True)
    for k in ("points", "normals", "curvature", "omega_p", "omega_n", "eigenvalues", "stats", "npoints"):
        neq = (o[k].reshape(len(o[k]), -1).view(np.uint32) != g[k].reshape(len(g[k]), -1).view(np.uint32)).any(1)
        md = np.abs(o[k].astype(np.float64) - g[k].astype(np.float64)).max() if neq.any() else 0.0
        print(f"{tag} {k:12s} rows differing {int(neq.sum()):7d} / {len(neq)}  max abs diff {md:.3e}")
    bad = np.nonzero((o["normals"].view(np.uint32) != g["normals"].view(np.uint32)).any(1))[0][:5]
    for i in bad:
        print("   idx", i, "o.n", o["normals"][i], "g.n", g["normals"][i], "o.ev", o["eigenvalues"][i], "g.ev", g["eigenvalues"][i])
o = O.align(ap, oc["ref"], oc["cur"])
aligner.setReferenceCloud(gc["ref"]); aligner.setCurrentCloud(gc["cur"])
g = aligner.align()
aligner.setReferenceCloud(upload(ctx, oc["ref"])); aligner.setCurrentCloud(upload(ctx, oc["cur"]))
g2 = aligner.align()
for i, it in enumerate(o["iterations"]):
    print(i, "oracle", it["chi2_fp64"], "gpu-conv", float(g["chi2"][i]), "rel %.2e" % (abs(g["chi2"][i] - it["chi2_fp64"]) / it["chi2_fp64"]),
          "gpu-upl", float(g2["chi2"][i]), "rel %.2e" % (abs(g2["chi2"][i] - it["chi2_fp64"]) / it["chi2_fp64"]), "C", it["C"], g["C"][i], g2["C"][i])
