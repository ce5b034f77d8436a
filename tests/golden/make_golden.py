#!/usr/bin/env python3
"""Generates the golden fixtures of tests/golden/ from the CPU oracle.

The reference holds no golden vectors for this path and cannot be run here, so these
fixtures freeze the ORACLE's outputs (PARITY UNPINNED: they pin regressions of the restatement and give the
GPU tests committed expected values; they are not outputs of the reference binary).

Run from the repo root:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import case_params, make_depth_pair  # noqa: E402
from oracle import oracle as O  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def canon(a):
    """-0.0 -> +0.0 so that hashes do not depend on the sign of zero."""
    a = np.array(a, copy=True)
    if a.dtype.kind == "f":
        a[a == 0] = 0
    return a


def hexf(a):
    return [format(int(x), "08x") for x in np.ascontiguousarray(a, np.float32).view(np.uint32).ravel()]


def cloud_record(cp, depth):
    cloud, idx, itv = O.convert(cp, depth)
    a = cloud.arrays(stats=True)
    pts, _ = O.unproject(cp, depth)
    I = O.integral_image(idx, pts)
    rec = dict(M=len(cloud), index_sha256=sha(idx), interval_sha256=sha(itv), integral_sha256=sha(canon(I)),
               first_points=hexf(a["points"][:4]), last_points=hexf(a["points"][-4:]))
    for k in ("points", "normals", "curvature", "omega_p", "omega_n", "eigenvalues", "npoints"):
        rec[k + "_sha256"] = sha(canon(a[k]))
    rec["valid_normals"] = int((np.abs(a["normals"][:, :3]).sum(1) > 0).sum())
    probes = np.linspace(0, len(cloud) - 1, 16).astype(int)
    rec["probe_index"] = probes.tolist()
    rec["probe_normals"] = hexf(a["normals"][probes]); rec["probe_curvature"] = hexf(a["curvature"][probes])
    return cloud, rec


def make(name, seed):
    rows, cols, K, conv, alig = case_params(name)
    ref, cur, Ttrue, ref_mm, cur_mm = make_depth_pair(name, seed)
    cp = O.converter_params(K=K, **conv)
    out = dict(case=name, seed=seed, rows=rows, cols=cols, K=list(K), converter=conv, aligner=alig,
               depth_ref_sha256=sha(ref_mm), depth_cur_sha256=sha(cur_mm), true_T=Ttrue.tolist())
    cr, out["reference"] = cloud_record(cp, ref)
    cc, out["current"] = cloud_record(cp, cur)
    pi, pd = O.project(K, np.eye(4), conv["min_distance"], conv["max_distance"], rows, cols, cr.arrays()["points"])
    out["project_identity"] = dict(index_sha256=sha(pi), depth_sha256=sha(pd))
    for mode in (0, 1):
        ap = O.aligner_params(rows, cols, K=K, accumulate_fp64=mode, **alig)
        r = O.align(ap, cr, cc, images=True)
        out["align_fp64" if mode else "align_fp32_serial"] = dict(
            T=hexf(r["T"]), error=hexf([r["error"]])[0], inliers=r["inliers"],
            iterations=[dict(K=it["K"], C=it["C"], inliers=it["inliers"], chi2=hexf([it["chi2"]])[0], chi2_fp64=it["chi2_fp64"],
                             T_before=hexf(it["T_before"])) for it in r["iterations"]],
            cur_index_sha256=sha(r["cur_index"]), cur_depth_sha256=sha(r["cur_depth"]), ref_index_sha256=sha(r["ref_index"]))
    with open(os.path.join(HERE, f"pwn_{name}_seed{seed}.json"), "w") as f:
        json.dump(out, f, indent=1)
    if name == "small":
        np.savez_compressed(os.path.join(HERE, f"pwn_{name}_seed{seed}_depth.npz"), ref_mm=ref_mm, cur_mm=cur_mm)
    print(name, seed, "M", out["reference"]["M"], out["current"]["M"], "chi2", [it["chi2_fp64"] for it in r["iterations"]][::3])


if __name__ == "__main__":
    make("small", 1)
    make("vga", 0)
