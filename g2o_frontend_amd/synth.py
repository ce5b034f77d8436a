"""Deterministic synthetic depth data for the PWN path.

The reference ships no depth images (``datasets/`` holds 2-D g2o graphs only), so every test, fixture
and benchmark in this repo runs on the scene defined here: a ray-cast "room" (floor y=+1.2, ceiling
y=-1.2, walls x=+-2.0, back wall z=4.0, camera looking down +z with image x right / y down) plus two
spheres of radius 0.4 m whose centres are drawn from the seed.  Depth is the camera-frame z
component, quantised to uint16 millimetres (what a Kinect driver delivers and what the reference's
``DepthImage_convert_16UC1_to_32FC1`` -- pwn_core/pwn_static.cpp:54-68 -- consumes), with 3 % of the
pixels zeroed by a counter-based hash.

Everything is float64 numpy arithmetic on correctly-rounded operations (+ - * / sqrt), so the same
seed gives the same uint16 image on every machine.
"""
from __future__ import annotations

import numpy as np

MASK64 = (1 << 64) - 1

# pwn_core/pwn_simple_aligner.cpp:226-229 (Kinect VGA) and SURVEY.md §8(d) config 5 (1280x960)
K_VGA = (525.0, 525.0, 319.5, 239.5)
K_1280 = (1050.0, 1050.0, 639.5, 479.5)


def splitmix64(x: np.ndarray | int) -> np.ndarray:
    """Vectorised splitmix64 finaliser on uint64."""
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def _uniform(seed: int, stream: int, n: int) -> np.ndarray:
    """n doubles in [0,1) from (seed, stream)."""
    base = (int(seed) * 0x632BE59BD9B4E019 + int(stream) * 0xD1342543DE82EF95) & MASK64
    ctr = (np.arange(n, dtype=np.uint64) + np.uint64(base))
    return (splitmix64(ctr) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def quat2mat(q: np.ndarray) -> np.ndarray:
    """Rotation from the vector part of a unit quaternion (pwn_core/bm_se3.h:9-22), float64."""
    qx, qy, qz = (float(v) for v in q)
    qw = np.sqrt(1.0 - (qx * qx + qy * qy + qz * qz))
    return np.array([
        [qw * qw + qx * qx - qy * qy - qz * qz, 2 * (qx * qy - qw * qz), 2 * (qx * qz + qw * qy)],
        [2 * (qx * qy + qz * qw), qw * qw - qx * qx + qy * qy - qz * qz, 2 * (qy * qz - qx * qw)],
        [2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), qw * qw - qx * qx - qy * qy + qz * qz]])


def v2t(v: np.ndarray) -> np.ndarray:
    """6-vector (t, q_xyz) -> 4x4 isometry, float64 (pwn_core/bm_se3.h:37-43)."""
    T = np.eye(4)
    T[:3, :3] = quat2mat(np.asarray(v[3:6], dtype=np.float64))
    T[:3, 3] = np.asarray(v[0:3], dtype=np.float64)
    return T


def scene_spheres(seed: int) -> np.ndarray:
    """Two sphere centres (2x3) inside the room at z in [1.5, 3]."""
    u = _uniform(seed, 1, 6)
    c = np.empty((2, 3))
    c[:, 0] = -1.2 + 2.4 * u[0:2]
    c[:, 1] = -0.5 + 1.2 * u[2:4]
    c[:, 2] = 1.5 + 1.5 * u[4:6]
    return c


def pair_pose(seed: int, t_max: float = 0.05, q_max: float = 0.02) -> np.ndarray:
    """Pose of the *current* camera in the reference camera frame: the transform Aligner::T() should
    recover (p_ref = T * p_cur).  t ~ U(+-t_max)^3, q ~ U(+-q_max)^3."""
    u = _uniform(seed, 2, 6)
    v = np.concatenate([(2 * u[0:3] - 1) * t_max, (2 * u[3:6] - 1) * q_max])
    return v2t(v)


def render_depth_mm(seed: int, pose: np.ndarray | None = None, rows: int = 480, cols: int = 640,
                    K=K_VGA, holes: float = 0.03, hole_stream: int = 0, noise: bool = False) -> np.ndarray:
    """Ray-cast the seeded scene from camera pose `pose` (camera-to-world 4x4); uint16 millimetres."""
    fx, fy, cx, cy = K
    if pose is None:
        pose = np.eye(4)
    R = np.asarray(pose[:3, :3], dtype=np.float64)
    o = np.asarray(pose[:3, 3], dtype=np.float64)
    v, u = np.meshgrid(np.arange(rows, dtype=np.float64), np.arange(cols, dtype=np.float64), indexing="ij")
    xn = (u - cx) / fx
    yn = (v - cy) / fy
    d = np.stack([R[i, 0] * xn + R[i, 1] * yn + R[i, 2] for i in range(3)], axis=0)   # 3 x rows x cols
    best = np.full((rows, cols), np.inf)

    def plane(axis: int, value: float):
        nonlocal best
        with np.errstate(divide="ignore", invalid="ignore"):
            s = (value - o[axis]) / d[axis]
        s = np.where(np.isfinite(s) & (s > 1e-9), s, np.inf)
        best = np.minimum(best, s)

    plane(1, 1.2); plane(1, -1.2); plane(0, 2.0); plane(0, -2.0); plane(2, 4.0)
    a = d[0] * d[0] + d[1] * d[1] + d[2] * d[2]
    for c in scene_spheres(seed):
        oc = o - c
        b = 2.0 * (d[0] * oc[0] + d[1] * oc[1] + d[2] * oc[2])
        cc = oc @ oc - 0.4 * 0.4
        disc = b * b - 4.0 * a * cc
        with np.errstate(invalid="ignore"):
            s = (-b - np.sqrt(np.where(disc >= 0, disc, 0.0))) / (2.0 * a)
        s = np.where((disc >= 0) & (s > 1e-9), s, np.inf)
        best = np.minimum(best, s)
    z = best                      # camera-frame z of the hit point (ray has unit z in the camera frame)
    if noise:
        un = _uniform(seed, 1000 + hole_stream, 2 * rows * cols).reshape(2, rows, cols)
        g = np.sqrt(-2.0 * np.log(1.0 - un[0])) * np.cos(2.0 * np.pi * un[1])
        z = z + g * (0.0012 + 0.0019 * (z - 0.4) ** 2)
    mm = np.floor(np.where(np.isfinite(z), z, 0.0) * 1000.0 + 0.5)
    mm = np.clip(mm, 0, 65535).astype(np.uint16)
    if holes > 0:
        base = (int(seed) * 0x9E3779B97F4A7C15 + int(hole_stream) * 0xC2B2AE3D27D4EB4F) & MASK64
        h = splitmix64(np.arange(rows * cols, dtype=np.uint64) + np.uint64(base)).reshape(rows, cols)
        mm[(h % np.uint64(10000)) < np.uint64(int(holes * 10000))] = 0
    return mm


def make_pair(seed: int, rows: int = 480, cols: int = 640, K=K_VGA, holes: float = 0.03, noise: bool = False):
    """(reference depth mm, current depth mm, true T[4x4 float64]) of the seeded pair; noise: the Kinect-like z-noise of
    render_depth_mm (sigma = 1.2 mm + 1.9 mm/m^2 (z - 0.4 m)^2), off for the parity fixtures."""
    T = pair_pose(seed)
    ref = render_depth_mm(seed, np.eye(4), rows, cols, K, holes, hole_stream=0, noise=noise)
    cur = render_depth_mm(seed, T, rows, cols, K, holes, hole_stream=1, noise=noise)
    return ref, cur, T


def trajectory(seed: int, n_frames: int, t_step: float = 0.02, r_step_deg: float = 1.0):
    """Smooth seeded camera trajectory for the sequential-odometry configuration (SURVEY.md §8(d)
    config 3): per-frame motion <= t_step metres and <= r_step_deg degrees."""
    u = _uniform(seed, 3, 12)
    poses = []
    T = np.eye(4)
    q_step = np.sin(np.deg2rad(r_step_deg) / 2.0)
    for k in range(n_frames):
        poses.append(T.copy())
        ph = 2.0 * np.pi * (k / 40.0)
        dv = np.array([
            0.6 * t_step * np.sin(ph + 6.28 * u[0]), 0.3 * t_step * np.sin(0.7 * ph + 6.28 * u[1]),
            0.5 * t_step * np.cos(0.5 * ph + 6.28 * u[2]),
            0.4 * q_step * np.sin(0.9 * ph + 6.28 * u[3]), 0.6 * q_step * np.cos(0.6 * ph + 6.28 * u[4]),
            0.3 * q_step * np.sin(0.8 * ph + 6.28 * u[5])])
        T = T @ v2t(dv)
    return poses


def scaled_K(K, scale: int):
    """Camera parameters after the harness' imageScale division (pwn_simple_aligner.cpp:144-147:
    the whole matrix, cx/cy included, is multiplied by 1/scale, K(2,2) reset to 1)."""
    inv = np.float32(1.0) / np.float32(scale)
    return tuple(float(np.float32(k) * inv) for k in K)


def K_matrix_colmajor(K) -> np.ndarray:
    """(fx,fy,cx,cy) -> column-major float32[9] camera matrix."""
    fx, fy, cx, cy = K
    return np.array([fx, 0, 0, 0, fy, 0, cx, cy, 1], dtype=np.float32)


def trajectory_sweep(seed: int, n_frames: int, yaw_deg: float = 30.0, t_step: float = 0.02, r_step_deg: float = 1.0):
    """Smooth seeded trajectory for BASELINE configs[2] that makes PwnTracker switch key-clouds (pwn_tracker/pwn_tracker.cpp:164-185):
    the camera pans left and right by +-yaw_deg (one full period over the stream) while swaying a few centimetres, so the overlap with the
    key-cloud falls below the new-frame fraction several times.  Per-frame motion stays <= t_step metres and <= r_step_deg degrees
    (SURVEY.md §8(d) config 3); returned poses are camera-to-world, pose[0] = identity."""
    u = _uniform(seed, 4, 8)
    n = max(int(n_frames), 2)
    A = np.deg2rad(yaw_deg)
    w = 2.0 * np.pi / n
    # the yaw rate A*w and the sway rate stay inside the per-frame limits whatever n is
    A = min(A, 0.9 * np.deg2rad(r_step_deg) / w)
    amp_t = min(0.25, 0.45 * t_step / w)
    poses = []
    for k in range(n_frames):
        ph = w * k
        yaw = A * np.sin(ph)
        pitch = np.deg2rad(1.5) * np.sin(2.0 * ph + 6.28 * u[0]) - np.deg2rad(1.5) * np.sin(6.28 * u[0])
        x = amp_t * (np.sin(ph + 6.28 * u[1]) - np.sin(6.28 * u[1]))
        z = 0.6 * amp_t * (np.sin(2.0 * ph + 6.28 * u[2]) - np.sin(6.28 * u[2]))
        y = 0.2 * amp_t * (np.sin(ph + 6.28 * u[3]) - np.sin(6.28 * u[3]))
        cy_, sy_ = np.cos(yaw), np.sin(yaw)
        cp_, sp_ = np.cos(pitch), np.sin(pitch)
        Ry = np.array([[cy_, 0, sy_], [0, 1, 0], [-sy_, 0, cy_]])
        Rx = np.array([[1, 0, 0], [0, cp_, -sp_], [0, sp_, cp_]])
        T = np.eye(4)
        T[:3, :3] = Ry @ Rx
        T[:3, 3] = (x, y, z)
        poses.append(T)
    return poses
