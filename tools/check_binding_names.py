#!/usr/bin/env python3
"""Syntax-level check of the reference-side binding (bindings/pwn_hip/*.{h,cpp}), which cannot be compiled in this image (it
includes the reference's headers: Eigen3 + OpenCV).  Every member variable (`_name`) and every method called through `->` or `.`
that the binding uses, and that the binding itself, the C-ABI, the standard library, Eigen or OpenCV do not declare, must be
declared in the reference headers the binding includes.  A miss = a typo or a stale name: exit code 1.  This is a name check --
explicitly NOT parity evidence and not a substitute for compiling the binding in the reference tree.

  tools/check_binding_names.py [/root/reference]"""
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_HEADERS = ["aligner", "linearizer", "correspondencefinder", "depthimageconverter", "depthimageconverterintegralimage", "cloud", "stats",
               "informationmatrix", "informationmatrixcalculator", "statscalculator", "statscalculatorintegralimage", "se3_prior", "pinholepointprojector",
               "pointprojector", "homogeneousvector4f", "pwn_typedefs", "gaussian3", "pointintegralimage", "pointaccumulator"]
# names that come from elsewhere: the C++ standard library, Eigen, OpenCV (cv::Mat), the C-ABI structs of include/pwn_hip.h
FOREIGN = set("""size resize empty push_back begin end first second find erase insert clear data c_str what str count at swap reserve front back
                 matrix block row col transpose inverse linear translation setIdentity setZero cast array diagonal norm normalized
                 create rows cols ptr total release clone type isContinuous
                 str what""".split())


def strip_comments(t):
    t = re.sub(r"/\*.*?\*/", " ", t, flags=re.S)
    t = re.sub(r"//[^\n]*", " ", t)
    return re.sub(r'"(\\.|[^"\\])*"', '""', t)


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    hdr_dir = os.path.join(ref, "g2o_frontend", "pwn_core")
    if not os.path.isdir(hdr_dir):
        print("reference headers not found under", hdr_dir); return 2
    ref_text = ""
    for h in REF_HEADERS:
        p = os.path.join(hdr_dir, h + ".h")
        if os.path.exists(p):
            ref_text += strip_comments(open(p, errors="replace").read()) + "\n"
    ref_text += strip_comments(open(os.path.join(ref, "g2o_frontend", "basemath", "bm_se3.h"), errors="replace").read()) if os.path.exists(os.path.join(ref, "g2o_frontend", "basemath", "bm_se3.h")) else ""
    capi = strip_comments(open(os.path.join(ROOT, "include", "pwn_hip.h")).read())
    files = sorted(glob.glob(os.path.join(ROOT, "bindings", "pwn_hip", "*.h")) + glob.glob(os.path.join(ROOT, "bindings", "pwn_hip", "*.cpp")))
    own = "\n".join(strip_comments(open(f).read()) for f in files)
    own_headers = "\n".join(strip_comments(open(f).read()) for f in files if f.endswith(".h"))
    # what the binding declares itself: members and methods in its own headers (class bodies) and functions it defines
    own_decl = set(re.findall(r"\b([A-Za-z_]\w*)\s*\(", own_headers)) | set(re.findall(r"\b(_[A-Za-z]\w*)\s*[;=,)\[{]", own_headers))
    own_decl |= set(re.findall(r"\b\w+::(\w+)\s*\(", own))
    capi_names = set(re.findall(r"\b([A-Za-z_]\w*)\b", capi))
    misses = []
    checked = {"members": set(), "methods": set()}
    for f in files:
        t = strip_comments(open(f).read())
        for m in sorted(set(re.findall(r"(?<!\w)(_[a-z][A-Za-z0-9]*)\b", t))):
            if m in own_decl:
                continue
            checked["members"].add(m)
            if not re.search(r"\b%s\b" % re.escape(m), ref_text):
                misses.append((os.path.basename(f), "member", m))
        for m in sorted(set(re.findall(r"(?:->|\.)\s*([A-Za-z_]\w*)\s*\(", t))):
            if m in own_decl or m in FOREIGN or m in capi_names:
                continue
            checked["methods"].add(m)
            if not re.search(r"\b%s\s*\(" % re.escape(m), ref_text):
                misses.append((os.path.basename(f), "method", m))
    print("binding name check: %d members, %d methods looked up in %d reference headers" % (len(checked["members"]), len(checked["methods"]), len(REF_HEADERS)))
    for f, kind, m in misses:
        print("  MISSING %s %s used in %s" % (kind, m, f))
    return 1 if misses else 0


if __name__ == "__main__":
    sys.exit(main())
