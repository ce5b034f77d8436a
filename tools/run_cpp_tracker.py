#!/usr/bin/env python3
"""BASELINE configs[2] through the C++ host mirror: writes the 200-frame synthetic VGA stream of bench.py's tracker leg as 16-bit PGM files and runs
tools/pwn_hip_tracker_app on it (matcher scale 1, the VGA configuration), one call after the other and with `lookAhead 1`; the app prints its frames/s
(tracking loop only, the track file's fprintf included).  usage: python tools/run_cpp_tracker.py [frames=200]"""
import concurrent.futures as cf
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONF = """depthScale 0.001
imageScale 1
fx 525.0
fy 525.0
cx 319.5
cy 239.5
minDistance 0.5
maxDistance 4.5
minImageRadius 10
maxImageRadius 30
minPoints 50
curvatureThreshold 0.2
worldRadius 0.1
informationMatrixCurvatureThreshold 0.02
inlierDistanceThreshold 1.0
inlierNormalAngularThreshold 0.95
inlierCurvatureRatioThreshold 1.3
flatCurvatureThreshold 0.02
inlierMaxChi2 9000
robustKernel 1
outerIterations 10
innerIterations 1
newFrameInliersFraction 0.4
cacheSize 2
"""


def main():
    from g2o_frontend_amd import build, synth
    build.build(); build.build_tools()
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    poses = synth.trajectory_sweep(9, n)
    with cf.ThreadPoolExecutor(16) as ex:
        frames = list(ex.map(lambda k: synth.render_depth_mm(9, poses[k], 480, 640, synth.K_VGA, hole_stream=k), range(n)))
    d = tempfile.mkdtemp(prefix="pwn_trk_")
    lst = []
    for k, f in enumerate(frames):
        fn = os.path.join(d, f"d{k}.pgm")
        with open(fn, "wb") as fh:
            fh.write(b"P5\n%d %d\n65535\n" % (f.shape[1], f.shape[0])); fh.write(f.astype(">u2").tobytes())
        lst.append(f"{k * 0.033:.3f} {fn}")
    open(os.path.join(d, "list.txt"), "w").write("\n".join(lst) + "\n")
    exe = os.path.join(ROOT, "tools", "pwn_hip_tracker_app")
    tracks = []
    for rep in range(2):
        for ahead in (0, 1):
            open(os.path.join(d, "conf.txt"), "w").write(CONF + f"lookAhead {ahead}\nwarmUp 1\n")
            prefix = os.path.join(d, f"run{ahead}")
            r = subprocess.run([exe, os.path.join(d, "conf.txt"), os.path.join(d, "list.txt"), prefix], capture_output=True, text=True, timeout=600)
            line = [l for l in r.stderr.splitlines() if l.startswith("tracking:")]
            print(line[0] if line else r.stderr[-300:], flush=True)
            tracks.append(open(prefix + "_track.txt", "rb").read())
    print("track files identical:", all(t == tracks[0] for t in tracks))


if __name__ == "__main__":
    main()
