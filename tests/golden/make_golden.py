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
    out = record(name, seed, rows, cols, K, conv, alig, ref_mm, cur_mm, Ttrue.tolist())
    with open(os.path.join(HERE, f"pwn_{name}_seed{seed}.json"), "w") as f:
        json.dump(out, f, indent=1)
    if name == "small":
        np.savez_compressed(os.path.join(HERE, f"pwn_{name}_seed{seed}_depth.npz"), ref_mm=ref_mm, cur_mm=cur_mm)


def record(name, seed, rows, cols, K, conv, alig, ref_mm, cur_mm, true_T):
    """Oracle outputs of one depth pair (uint16 millimetre frames) as a JSON-able record."""
    ref, cur = O.convert_16u_to_32f(ref_mm), O.convert_16u_to_32f(cur_mm)
    cp = O.converter_params(K=K, **conv)
    out = dict(case=name, seed=seed, rows=rows, cols=cols, K=list(K), converter=conv, aligner=alig,
               depth_ref_sha256=sha(ref_mm), depth_cur_sha256=sha(cur_mm), true_T=true_T)
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
                             T_before=hexf(it["T_before"]), H=hexf(it["H"]), b=hexf(it["b"])) for it in r["iterations"]],
            cur_index_sha256=sha(r["cur_index"]), cur_depth_sha256=sha(r["cur_depth"]), ref_index_sha256=sha(r["ref_index"]))
    print(name, seed, "M", out["reference"]["M"], out["current"]["M"], "chi2", [it["chi2_fp64"] for it in r["iterations"]][::3])
    return out


# ---- real sensor data -------------------------------------------------------------------------------------------
# The reference repository holds five 640x480 16-bit depth frames of a Kinect (g2o_frontend/PlaneEx_gui/test_images/*.pgm,
# millimetres, the input format of pwn_simple_aligner.cpp:137); image_OLD_1 / image_OLD_2 are two views of the same scene about
# 8 mm / 1.5 degrees apart.  They are DATA (inputs only: the reference holds no expected outputs for them), copied here as a
# compressed array so that the parity tests also run on real sensor noise, holes and non-flat surfaces (25 000 points of
# image_OLD_1 take the 1/lambda branch of informationmatrixcalculator.cpp:27-29, which the synthetic room hardly reaches).
REAL_DIR = "/root/reference/g2o_frontend/PlaneEx_gui/test_images"
REAL_PAIR = ("image_OLD_1.pgm", "image_OLD_2.pgm")


def read_pgm16(path):
    import re
    b = open(path, "rb").read()
    m = re.match(rb"P5\s+(\d+)\s+(\d+)\s+(\d+)\s", b)
    w, h, mx = map(int, m.groups())
    assert mx == 65535
    return np.frombuffer(b[m.end():m.end() + 2 * w * h], dtype=">u2").reshape(h, w).astype(np.uint16)


def make_real():
    npz = os.path.join(HERE, "kinect_real_pair.npz")
    if os.path.isdir(REAL_DIR):
        ref_mm, cur_mm = (read_pgm16(os.path.join(REAL_DIR, f)) for f in REAL_PAIR)
        np.savez_compressed(npz, ref_mm=ref_mm, cur_mm=cur_mm)
    z = np.load(npz)
    rows, cols, K, conv, alig = case_params("vga")
    out = record("kinect", 0, rows, cols, K, conv, alig, z["ref_mm"], z["cur_mm"], None)
    with open(os.path.join(HERE, "pwn_kinect_seed0.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    make("small", 1)
    make("vga", 0)
    make_real()
