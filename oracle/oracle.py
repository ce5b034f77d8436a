"""ctypes wrapper of the CPU oracle (oracle/pwn_oracle.{h,cpp}).

TEST INFRASTRUCTURE ONLY: may be imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product package (g2o_frontend_amd).  PARITY UNPINNED (see
pwn_oracle.h): the oracle restates the reference; no reference binary or golden vector exists.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpwn_oracle.so")


def build(force: bool = False) -> str:
    src = [os.path.join(_HERE, f) for f in ("pwn_oracle.cpp", "pwn_oracle.h", "Makefile")]
    stale = (not os.path.exists(_LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in src)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libpwn_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def _cpu_tag() -> str:
    """identifies the host CPU a -march=native build belongs to (model name + instruction-set flags)"""
    import hashlib
    model, flags = "unknown", ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "unknown":
                model = line.split(":", 1)[1].strip()
            elif line.startswith("flags") and not flags:
                flags = line.split(":", 1)[1].strip()
    except OSError:
        pass
    return hashlib.sha1((model + "|" + flags).encode()).hexdigest()[:12]


def build_fast(force: bool = False) -> str:
    """The timed CPU baseline's build of the same source: -O3 -march=native (still -ffp-contract=off, no fast-math: same results).
    -march=native binds the binary to the CPU it was compiled on, so it is built on the machine that runs it (the GPU box's host),
    into oracle/_fast/<cpu tag>/ (BASELINE.md section 3)."""
    out_dir = os.path.join(_HERE, "_fast", _cpu_tag())
    out = os.path.join(out_dir, "libpwn_oracle.so")
    src = [os.path.join(_HERE, f) for f in ("pwn_oracle.cpp", "pwn_oracle.h")]
    stale = (not os.path.exists(out)) or any(os.path.getmtime(s) > os.path.getmtime(out) for s in src)
    if force or stale:
        os.makedirs(out_dir, exist_ok=True)
        subprocess.check_call([os.environ.get("CXX", "g++"), "-O3", "-march=native", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
                               "-fopenmp", "-shared", "-o", out, src[0]])
    return out


class ConverterParams(C.Structure):
    _fields_ = [("K", C.c_float * 9), ("min_distance", C.c_float), ("max_distance", C.c_float),
                ("world_radius", C.c_float), ("min_image_radius", C.c_int), ("max_image_radius", C.c_int),
                ("min_points", C.c_int), ("stats_curvature_threshold", C.c_float),
                ("point_info_curvature_threshold", C.c_float), ("normal_info_curvature_threshold", C.c_float),
                ("point_flat_diag", C.c_float * 3), ("point_nonflat_diag", C.c_float * 3),
                ("normal_flat_diag", C.c_float * 3), ("normal_nonflat_diag", C.c_float * 3),
                ("sensor_offset", C.c_float * 16)]


class AlignerParams(C.Structure):
    _fields_ = [("K", C.c_float * 9), ("min_distance", C.c_float), ("max_distance", C.c_float),
                ("rows", C.c_int), ("cols", C.c_int), ("inlier_distance_threshold", C.c_float),
                ("inlier_normal_angular_threshold", C.c_float), ("flat_curvature_threshold", C.c_float),
                ("inlier_curvature_ratio_threshold", C.c_float), ("inlier_max_chi2", C.c_float),
                ("robust_kernel", C.c_int), ("outer_iterations", C.c_int), ("inner_iterations", C.c_int),
                ("reference_sensor_offset", C.c_float * 16), ("current_sensor_offset", C.c_float * 16),
                ("initial_guess", C.c_float * 16), ("accumulate_fp64", C.c_int)]


class IterTrace(C.Structure):
    _fields_ = [("K", C.c_int), ("C", C.c_int), ("inliers", C.c_int), ("chi2", C.c_float),
                ("chi2_fp64", C.c_double), ("H", C.c_float * 36), ("b", C.c_float * 6),
                ("T_before", C.c_float * 16)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        # PWN_ORACLE_VARIANT=fast (bench.py's cpu_baseline child only): the -O3 -march=native build of the same source
        path = None
        if os.environ.get("PWN_ORACLE_VARIANT") == "fast":
            try:
                path = build_fast()
            except Exception:      # no compiler / read-only tree on this host: time the checker build instead (same results, -O2)
                path = None
        if path is None:
            path = build()
        L = C.CDLL(path)
        L._pwn_variant = "fast" if path != _LIB_PATH else "checker"
        L.orc_cloud_create.restype = C.c_void_p
        L.orc_cloud_destroy.argtypes = [C.c_void_p]
        L.orc_cloud_size.argtypes = [C.c_void_p]
        L.orc_cloud_size.restype = C.c_int
        L.orc_cloud_get.argtypes = [C.c_void_p] + [C.c_void_p] * 8
        L.orc_cloud_set.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 5
        L.orc_unproject.restype = C.c_int
        L.orc_unproject.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_project_intervals.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_integral_image.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_convert.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_project.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                  C.c_void_p, C.c_void_p]
        L.orc_correspondences.restype = C.c_int
        L.orc_correspondences.argtypes = [C.c_void_p] * 8
        L.orc_linearize.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 6
        L.orc_align.argtypes = [C.c_void_p] * 11
        L.orc_convert_16u_to_32f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float]
        L.orc_convert_32f_to_16u.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float]
        L.orc_depth_scale.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]
        L.orc_projector_matrices.argtypes = [C.c_void_p] * 5
        L.orc_v2t.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_t2v.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_eigen3.argtypes = [C.c_void_p] * 3
        L.orc_ldlt_solve6.argtypes = [C.c_void_p] * 3
        L.orc_num_threads.restype = C.c_int
        L.orc_set_num_threads.argtypes = [C.c_int]
        L.orc_set_parallel_align.argtypes = [C.c_int]
        L.orc_set_trig_mode.argtypes = [C.c_int]
        L.orc_trig_eval.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 5
        L.orc_add_prior.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_compute_statistics.argtypes = [C.c_void_p] * 6
        L.orc_align_statistics.argtypes = [C.c_void_p] * 9
        L.orc_iso_inverse.argtypes = [C.c_void_p] * 2
        L.orc_iso_mul.argtypes = [C.c_void_p] * 3
        L.orc_reorthonormalize.argtypes = [C.c_void_p] * 2
        L.orc_match_score.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float] + [C.c_void_p] * 4
        L.orc_set_gaussians.argtypes = [C.c_int, C.c_float, C.c_float]
        L.orc_cloud_num_gaussians.argtypes = [C.c_void_p]; L.orc_cloud_num_gaussians.restype = C.c_int
        L.orc_cloud_get_gaussians.argtypes = [C.c_void_p] * 6
        L.orc_cloud_transform_in_place.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_cloud_add.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_merge.restype = C.c_int
        L.orc_merge.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_void_p]
        L.orc_voxelize.restype = C.c_int
        L.orc_voxelize.argtypes = [C.c_void_p, C.c_float, C.c_int, C.c_void_p]
        L.orc_cloud_save.restype = C.c_int
        L.orc_cloud_save.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int, C.c_int]
        L.orc_cloud_load.restype = C.c_int
        L.orc_cloud_load.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _set_mat(field, M):
    """4x4 or 3x3 numpy (row-indexed) -> column-major ctypes float array."""
    flat = np.asarray(M, dtype=np.float32).T.reshape(-1)
    for i, v in enumerate(flat):
        field[i] = float(v)


def mat_from_colmajor(a, n=4):
    return np.array(list(a), dtype=np.float32).reshape(n, n).T.copy()


def converter_params(K=(525.0, 525.0, 319.5, 239.5), sensor_offset=None, **kw) -> ConverterParams:
    p = ConverterParams()
    lib().orc_default_converter_params(C.byref(p))
    fx, fy, cx, cy = K
    for i, v in enumerate([fx, 0, 0, 0, fy, 0, cx, cy, 1]):
        p.K[i] = v
    if sensor_offset is not None:
        _set_mat(p.sensor_offset, sensor_offset)
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        if isinstance(v, (tuple, list, np.ndarray)):
            for i, x in enumerate(v):
                getattr(p, k)[i] = float(x)
        else:
            setattr(p, k, v)
    return p


def aligner_params(rows, cols, K=(525.0, 525.0, 319.5, 239.5), initial_guess=None, reference_sensor_offset=None,
                   current_sensor_offset=None, **kw) -> AlignerParams:
    p = AlignerParams()
    lib().orc_default_aligner_params(C.byref(p))
    fx, fy, cx, cy = K
    for i, v in enumerate([fx, 0, 0, 0, fy, 0, cx, cy, 1]):
        p.K[i] = v
    p.rows, p.cols = rows, cols
    if initial_guess is not None:
        _set_mat(p.initial_guess, initial_guess)
    if reference_sensor_offset is not None:
        _set_mat(p.reference_sensor_offset, reference_sensor_offset)
    if current_sensor_offset is not None:
        _set_mat(p.current_sensor_offset, current_sensor_offset)
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


# Parameter sets of the reference configs
VGA_CONF_CONVERTER = dict(min_distance=0.5, max_distance=4.5, world_radius=0.1, min_image_radius=10,
                          max_image_radius=30, min_points=50, stats_curvature_threshold=0.2,
                          point_info_curvature_threshold=0.02, normal_info_curvature_threshold=0.02)
VGA_CONF_ALIGNER = dict(min_distance=0.5, max_distance=4.5, inlier_distance_threshold=1.0,
                        inlier_normal_angular_threshold=0.95, flat_curvature_threshold=0.02,
                        inlier_curvature_ratio_threshold=1.3, inlier_max_chi2=9000.0, robust_kernel=1,
                        outer_iterations=10, inner_iterations=1)
# pwn_core/conf/pwn_aligner_1_4.conf (imageScale 4)
QVGA4_CONF_CONVERTER = dict(VGA_CONF_CONVERTER, min_image_radius=3, max_image_radius=6, min_points=10)
QVGA4_CONF_ALIGNER = dict(VGA_CONF_ALIGNER, inlier_distance_threshold=0.5)


class Cloud:
    """Host cloud owned by the oracle library."""

    def __init__(self):
        self.h = C.c_void_p(lib().orc_cloud_create())

    def __del__(self):
        try:
            if self.h:
                lib().orc_cloud_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def __len__(self):
        return lib().orc_cloud_size(self.h)

    def arrays(self, stats=False):
        n = len(self)
        out = dict(points=np.empty((n, 4), np.float32), normals=np.empty((n, 4), np.float32),
                   curvature=np.empty(n, np.float32), omega_p=np.empty((n, 16), np.float32),
                   omega_n=np.empty((n, 16), np.float32))
        st = ev = npts = None
        if stats:
            st = np.empty((n, 16), np.float32); ev = np.empty((n, 3), np.float32); npts = np.empty(n, np.int32)
            out.update(stats=st, eigenvalues=ev, npoints=npts)
        lib().orc_cloud_get(self.h, _p(out["points"]), _p(out["normals"]), _p(out["curvature"]), _p(st), _p(ev), _p(npts),
                            _p(out["omega_p"]), _p(out["omega_n"]))
        return out

    # ---- scene maintenance ----
    def num_gaussians(self):
        return lib().orc_cloud_num_gaussians(self.h)

    def gaussians(self):
        n = self.num_gaussians()
        out = dict(mean=np.empty((n, 3), np.float32), cov=np.empty((n, 9), np.float32), info_vec=np.empty((n, 3), np.float32),
                   info=np.empty((n, 9), np.float32), flags=np.empty(n, np.int32))
        lib().orc_cloud_get_gaussians(self.h, _p(out["mean"]), _p(out["cov"]), _p(out["info_vec"]), _p(out["info"]), _p(out["flags"]))
        return out

    def transform_in_place(self, T):
        Tc = _f32(np.asarray(T, np.float32).T.reshape(-1))
        lib().orc_cloud_transform_in_place(self.h, _p(Tc))

    def add(self, other: "Cloud", T=None):
        """Cloud::add (cloud.cpp:145-171)"""
        Tc = _f32(np.asarray(np.eye(4) if T is None else T, np.float32).T.reshape(-1))
        lib().orc_cloud_add(self.h, other.h, _p(Tc))

    def save(self, filename, T=None, step=1, binary=False):
        Tc = _f32(np.asarray(np.eye(4) if T is None else T, np.float32).T.reshape(-1))
        return bool(lib().orc_cloud_save(self.h, str(filename).encode(), _p(Tc), int(step), int(bool(binary))))

    @staticmethod
    def load(filename):
        c = Cloud(); T = np.empty(16, np.float32)
        ok = bool(lib().orc_cloud_load(c.h, str(filename).encode(), _p(T)))
        return (c if ok else None), T.reshape(4, 4).T.copy()

    @staticmethod
    def from_arrays(points, normals, curvature, omega_p, omega_n):
        c = Cloud()
        pts, nrm, cur, op, on = _f32(points), _f32(normals), _f32(curvature), _f32(omega_p), _f32(omega_n)
        lib().orc_cloud_set(c.h, len(pts), _p(pts), _p(nrm), _p(cur), _p(op), _p(on))
        return c


def convert_16u_to_32f(raw, scale=0.001):
    raw = np.ascontiguousarray(raw, dtype=np.uint16)
    out = np.empty(raw.shape, np.float32)
    lib().orc_convert_16u_to_32f(_p(raw), _p(out), raw.size, scale)
    return out


def convert_32f_to_16u(img, scale=1000.0):
    img = _f32(img)
    out = np.empty(img.shape, np.uint16)
    lib().orc_convert_32f_to_16u(_p(img), _p(out), img.size, scale)
    return out


def depth_scale(img, step, max_depth_cov=0.01):
    img = _f32(img)
    out = np.empty((img.shape[0] // step, img.shape[1] // step), np.float32)
    lib().orc_depth_scale(_p(img), img.shape[0], img.shape[1], step, max_depth_cov, _p(out))
    return out


def projector_matrices(K, T):
    Kc = np.array([K[0], 0, 0, 0, K[1], 0, K[2], K[3], 1], np.float32)
    Tc = _f32(np.asarray(T, np.float32).T.reshape(-1))
    KRt = np.empty(16, np.float32); iKRt = np.empty(16, np.float32); iK = np.empty(9, np.float32)
    lib().orc_projector_matrices(_p(Kc), _p(Tc), _p(KRt), _p(iKRt), _p(iK))
    return KRt.reshape(4, 4).T.copy(), iKRt.reshape(4, 4).T.copy(), iK.reshape(3, 3).T.copy()


def unproject(p: ConverterParams, depth):
    depth = _f32(depth); rows, cols = depth.shape
    pts = np.empty((rows * cols, 4), np.float32); idx = np.empty((rows, cols), np.int32)
    n = lib().orc_unproject(C.byref(p), _p(depth), rows, cols, _p(pts), _p(idx))
    return pts[:n].copy(), idx


def project_intervals(p: ConverterParams, depth):
    depth = _f32(depth); rows, cols = depth.shape
    out = np.empty((rows, cols), np.int32)
    lib().orc_project_intervals(C.byref(p), _p(depth), rows, cols, _p(out))
    return out


def integral_image(index_image, points):
    idx = np.ascontiguousarray(index_image, np.int32); pts = _f32(points)
    rows, cols = idx.shape
    out = np.empty((10, rows, cols), np.float32)
    lib().orc_integral_image(_p(idx), _p(pts), rows, cols, _p(out))
    return out


def convert(p: ConverterParams, depth):
    """DepthImageConverterIntegralImage::compute -> (Cloud, index image, interval image)."""
    depth = _f32(depth); rows, cols = depth.shape
    c = Cloud(); idx = np.empty((rows, cols), np.int32); itv = np.empty((rows, cols), np.int32)
    lib().orc_convert(C.byref(p), _p(depth), rows, cols, c.h, _p(idx), _p(itv))
    return c, idx, itv


def project(K, T, min_distance, max_distance, rows, cols, points):
    Kc = np.array([K[0], 0, 0, 0, K[1], 0, K[2], K[3], 1], np.float32)
    Tc = _f32(np.asarray(T, np.float32).T.reshape(-1))
    pts = _f32(points)
    idx = np.empty((rows, cols), np.int32); dep = np.empty((rows, cols), np.float32)
    lib().orc_project(_p(Kc), _p(Tc), min_distance, max_distance, rows, cols, _p(pts), len(pts), _p(idx), _p(dep))
    return idx, dep


def correspondences(p: AlignerParams, ref: Cloud, cur: Cloud, ref_index, cur_index, T):
    ri = np.ascontiguousarray(ref_index, np.int32); ci = np.ascontiguousarray(cur_index, np.int32)
    Tc = _f32(np.asarray(T, np.float32).T.reshape(-1))
    corr = np.empty((ri.size, 2), np.int32); K = C.c_int(0)
    n = lib().orc_correspondences(C.byref(p), ref.h, cur.h, _p(ri), _p(ci), _p(Tc), _p(corr), C.byref(K))
    return corr[:n].copy(), K.value


def linearize(p: AlignerParams, ref: Cloud, cur: Cloud, corr, T):
    corr = np.ascontiguousarray(corr, np.int32)
    Tc = _f32(np.asarray(T, np.float32).T.reshape(-1))
    H = np.empty(36, np.float32); b = np.empty(6, np.float32)
    chi2 = C.c_float(0); chi2d = C.c_double(0); inl = C.c_int(0)
    lib().orc_linearize(C.byref(p), ref.h, cur.h, _p(corr), len(corr), _p(Tc), _p(H), _p(b), C.byref(chi2),
                        C.byref(chi2d), C.byref(inl))
    return dict(H=H.reshape(6, 6).T.copy(), b=b, chi2=chi2.value, chi2_fp64=chi2d.value, inliers=inl.value)


def align(p: AlignerParams, ref: Cloud, cur: Cloud, images=False):
    n_it = p.outer_iterations * p.inner_iterations
    trace = (IterTrace * n_it)()
    T = np.empty(16, np.float32); err = C.c_float(0); inl = C.c_int(0)
    N = p.rows * p.cols
    ri = rd = ci = cd = None
    if images:
        ri = np.empty((p.rows, p.cols), np.int32); ci = np.empty((p.rows, p.cols), np.int32)
        rd = np.empty((p.rows, p.cols), np.float32); cd = np.empty((p.rows, p.cols), np.float32)
    lib().orc_align(C.byref(p), ref.h, cur.h, _p(T), C.byref(err), C.byref(inl), trace, _p(ri), _p(rd), _p(ci), _p(cd))
    its = [dict(K=t.K, C=t.C, inliers=t.inliers, chi2=t.chi2, chi2_fp64=t.chi2_fp64,
                H=np.array(list(t.H), np.float32).reshape(6, 6).T.copy(), b=np.array(list(t.b), np.float32),
                T_before=mat_from_colmajor(t.T_before)) for t in trace]
    out = dict(T=T.reshape(4, 4).T.copy(), error=err.value, inliers=inl.value, iterations=its)
    if images:
        out.update(ref_index=ri, ref_depth=rd, cur_index=ci, cur_depth=cd)
    return out


def set_gaussians(enabled: bool, baseline=0.075, alpha=0.1):
    """convert() also produces the sensor-noise Gaussians (pinholepointprojector.cpp:10-11 defaults)"""
    lib().orc_set_gaussians(1 if enabled else 0, baseline, alpha)


def merge(cloud: "Cloud", K, T, min_distance, max_distance, rows, cols, distance_threshold=0.1,
          normal_threshold=float(np.cos(np.float32(10 * np.pi / 180.0))), max_point_depth=10.0):
    """Merger::merge (merger.cpp:15-119; defaults :6-8) -> (new size, _collapsedIndices)"""
    Kc = np.array([K[0], 0, 0, 0, K[1], 0, K[2], K[3], 1], np.float32)
    Tc = _f32(np.asarray(T, np.float32).T.reshape(-1))
    collapsed = np.empty(len(cloud), np.int32)
    k = lib().orc_merge(cloud.h, _p(Kc), _p(Tc), min_distance, max_distance, rows, cols, distance_threshold, normal_threshold,
                        max_point_depth, _p(collapsed))
    return k, collapsed


def voxelize(cloud: "Cloud", resolution=0.01, literal=False):
    """VoxelCalculator::compute -> (new size, original indices of the survivors in output order)"""
    kept = np.empty(max(1, len(cloud)), np.int32)
    k = lib().orc_voxelize(cloud.h, resolution, 1 if literal else 0, _p(kept))
    return k, kept[:k].copy()


def match_score(ref_depth, cur_depth, threshold=50.0):
    """pwn_matcher_base.cpp:153-182 -> dict(image_nonZeros, image_outliers, image_inliers, image_reprojectionDistance)"""
    r, c = _f32(ref_depth), _f32(cur_depth)
    nz, out, inl, dist = C.c_int(0), C.c_int(0), C.c_int(0), C.c_float(0)
    lib().orc_match_score(_p(r), _p(c), r.size, C.c_float(threshold), C.byref(nz), C.byref(out), C.byref(inl), C.byref(dist))
    return dict(image_nonZeros=nz.value, image_outliers=out.value, image_inliers=inl.value, image_reprojectionDistance=dist.value)


def clear_priors():
    lib().orc_clear_priors()


def add_prior(kind, mean, information, reference_transform=None):
    m = _f32(np.asarray(mean, np.float32).T.reshape(-1))
    r = _f32(np.asarray(np.eye(4) if reference_transform is None else reference_transform, np.float32).T.reshape(-1))
    i = _f32(np.asarray(information, np.float32).T.reshape(-1))
    lib().orc_add_prior(int(kind), _p(m), _p(r), _p(i))


def compute_statistics(H, T):
    """aligner.cpp:152-199 on a given linearizer H (row-indexed 6x6) and final transform"""
    Hc = _f32(np.asarray(H, np.float32).T.reshape(-1)); Tc = _f32(np.asarray(T, np.float32).T.reshape(-1))
    mean = np.empty(6, np.float32); om = np.empty(36, np.float32); tr, rr = C.c_float(0), C.c_float(0)
    lib().orc_compute_statistics(_p(Hc), _p(Tc), _p(mean), _p(om), C.byref(tr), C.byref(rr))
    return dict(mean=mean, omega=om.reshape(6, 6).T.copy(), translationalEigenRatio=tr.value, rotationalEigenRatio=rr.value)


def align_statistics(p: AlignerParams, ref: "Cloud", cur: "Cloud", T):
    """_computeStatistics after the last align() of this process (11th linearizer update + statistics)"""
    Tc = _f32(np.asarray(T, np.float32).T.reshape(-1))
    H = np.empty(36, np.float32); mean = np.empty(6, np.float32); om = np.empty(36, np.float32); tr, rr = C.c_float(0), C.c_float(0)
    lib().orc_align_statistics(C.byref(p), ref.h, cur.h, _p(Tc), _p(H), _p(mean), _p(om), C.byref(tr), C.byref(rr))
    return dict(H=H.reshape(6, 6).T.copy(), mean=mean, omega=om.reshape(6, 6).T.copy(), translationalEigenRatio=tr.value, rotationalEigenRatio=rr.value)


def iso_inverse(T):
    Tc = _f32(np.asarray(T, np.float32).T.reshape(-1)); o = np.empty(16, np.float32)
    lib().orc_iso_inverse(_p(Tc), _p(o))
    return o.reshape(4, 4).T.copy()


def iso_mul(A, B):
    a = _f32(np.asarray(A, np.float32).T.reshape(-1)); b = _f32(np.asarray(B, np.float32).T.reshape(-1)); o = np.empty(16, np.float32)
    lib().orc_iso_mul(_p(a), _p(b), _p(o))
    return o.reshape(4, 4).T.copy()


def reorthonormalize(T):
    """pwn_tracker/pwn_tracker.cpp:154-159"""
    a = _f32(np.asarray(T, np.float32).T.reshape(-1)); o = np.empty(16, np.float32)
    lib().orc_reorthonormalize(_p(a), _p(o))
    return o.reshape(4, 4).T.copy()


def v2t(v):
    v = _f32(v); T = np.empty(16, np.float32)
    lib().orc_v2t(_p(v), _p(T))
    return T.reshape(4, 4).T.copy()


def t2v(T):
    Tc = _f32(np.asarray(T, np.float32).T.reshape(-1)); v = np.empty(6, np.float32)
    lib().orc_t2v(_p(Tc), _p(v))
    return v


def eigen3(A):
    Ac = _f32(np.asarray(A, np.float32).T.reshape(-1)); ev = np.empty(3, np.float32); U = np.empty(9, np.float32)
    lib().orc_eigen3(_p(Ac), _p(ev), _p(U))
    return ev, U.reshape(3, 3).T.copy()


def ldlt_solve6(H, b):
    Hc = _f32(np.asarray(H, np.float32).T.reshape(-1)); bc = _f32(b); x = np.empty(6, np.float32)
    lib().orc_ldlt_solve6(_p(Hc), _p(bc), _p(x))
    return x


def num_threads():
    return lib().orc_num_threads()


def set_trig_mode(mode):
    """0 / False: canonical fixed double algorithms; 1 / True: literal float libm; 2: double libm rounded to float"""
    lib().orc_set_trig_mode(int(mode))


def trig_eval(mode, y, x):
    """(theta, cos, sin) floats of the eigensolver for y = sqrt(q), x = half_b in trig mode `mode`"""
    y = _f32(y); x = _f32(x); n = y.size
    th = np.empty(n, np.float32); c = np.empty(n, np.float32); s = np.empty(n, np.float32)
    lib().orc_trig_eval(int(mode), n, _p(y), _p(x), _p(th), _p(c), _p(s))
    return th, c, s


def set_num_threads(n):
    lib().orc_set_num_threads(int(n))


def set_parallel_align(on: bool):
    """CorrespondenceFinder::compute / Linearizer::update over the OpenMP threads with the reference's partition minus its remainder
    dropping (correspondencefinder.cpp:38-51, linearizer.cpp:32-39).  Off (default) = canonical one-thread loops.  Timed baseline only."""
    lib().orc_set_parallel_align(1 if on else 0)
