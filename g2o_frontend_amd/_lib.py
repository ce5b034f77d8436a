"""Loader and ctypes prototypes of the C-ABI shared library (include/pwn_hip.h).

There is no CPU fallback: if ``libpwn_hip.so`` has not been built (``python -m g2o_frontend_amd.build``
or ``__graft_entry__.build()``) importing this module's ``lib()`` raises, and without a HIP device
``pwn_hip_ctx_create`` returns PWN_HIP_ERR_NO_DEVICE.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PWN_HIP_LIB", os.path.join(_HERE, "libpwn_hip.so"))   # override: kernel A/B experiments
MAX_ITERATIONS = 64

STATUS = {0: "OK", 1: "INVALID_ARGUMENT", 2: "NO_DEVICE", 3: "ALLOCATION", 4: "COPY", 5: "LAUNCH", 6: "CAPACITY"}


class PwnHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"pwn_hip status {code} ({STATUS.get(code, '?')}): {msg}")
        self.code = code


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
                ("initial_guess", C.c_float * 16)]


class AlignResult(C.Structure):
    _fields_ = [("T", C.c_float * 16), ("error", C.c_float), ("inliers", C.c_int), ("iterations", C.c_int),
                ("total_time_ms", C.c_float), ("chi2", C.c_float * MAX_ITERATIONS),
                ("iter_inliers", C.c_int * MAX_ITERATIONS), ("iter_correspondences", C.c_int * MAX_ITERATIONS),
                ("iter_candidates", C.c_int * MAX_ITERATIONS), ("n_reference", C.c_int), ("n_current", C.c_int)]


class MatchResult(C.Structure):
    _fields_ = [("image_non_zeros", C.c_int), ("image_outliers", C.c_int), ("image_inliers", C.c_int),
                ("image_reprojection_distance", C.c_float)]


class AlignStatistics(C.Structure):
    _fields_ = [("mean", C.c_float * 6), ("omega", C.c_float * 36), ("translational_eigen_ratio", C.c_float),
                ("rotational_eigen_ratio", C.c_float), ("H", C.c_float * 36), ("b", C.c_float * 6), ("error", C.c_float), ("inliers", C.c_int)]


class Prior(C.Structure):
    _fields_ = [("kind", C.c_int), ("mean", C.c_float * 16), ("reference_transform", C.c_float * 16), ("information", C.c_float * 36)]


# name -> (restype, argtypes); every symbol include/pwn_hip.h declares
_VP, _I, _F = C.c_void_p, C.c_int, C.c_float
PROTOTYPES = {
    "pwn_hip_ctx_create": (_I, [C.POINTER(_VP), _I, _I, _I, _I]),
    "pwn_hip_ctx_destroy": (_I, [_VP]),
    "pwn_hip_ctx_set_stream": (_I, [_VP, _VP]),
    "pwn_hip_ctx_synchronize": (_I, [_VP]),
    "pwn_hip_ctx_wait_stream": (_I, [_VP, _VP]),
    "pwn_hip_ctx_signal_stream": (_I, [_VP, _VP]),
    "pwn_hip_ctx_set_enqueued_callback": (_I, [_VP, _VP, _VP]),
    "pwn_hip_ctx_set_subbatch": (_I, [_VP, _I, _I]),
    "pwn_hip_ctx_set_concurrency": (_I, [_VP, _I]),
    "pwn_hip_ctx_set_omega_storage": (_I, [_VP, _I]),
    "pwn_hip_cloud_omega_storage": (_I, [_VP, _VP, C.POINTER(_I)]),
    "pwn_hip_last_error_string": (C.c_char_p, [_VP]),
    "pwn_hip_device_count": (_I, []),
    "pwn_hip_host_alloc": (_I, [C.POINTER(C.c_void_p), C.c_size_t]),
    "pwn_hip_host_free": (_I, [_VP]),
    "pwn_hip_device_alloc": (_I, [_VP, C.POINTER(C.c_void_p), C.c_size_t]),
    "pwn_hip_device_free": (_I, [_VP, _VP]),
    "pwn_hip_copy": (_I, [_VP, _VP, _VP, C.c_size_t]),
    "pwn_hip_copy_async": (_I, [_VP, _VP, _VP, C.c_size_t]),
    "pwn_hip_default_converter_params": (None, [_VP]),
    "pwn_hip_default_aligner_params": (None, [_VP]),
    "pwn_hip_cloud_create": (_I, [_VP, _I, C.POINTER(_VP)]),
    "pwn_hip_cloud_destroy": (_I, [_VP, _VP]),
    "pwn_hip_cloud_size": (_I, [_VP, _VP, C.POINTER(_I)]),
    "pwn_hip_cloud_upload": (_I, [_VP, _VP, _I, _VP, _VP, _VP, _VP, _VP]),
    "pwn_hip_cloud_download": (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "pwn_hip_cloud_download_stats": (_I, [_VP, _VP, _VP, _VP, _VP]),
    "pwn_hip_cloud_transform_in_place": (_I, [_VP, _VP, _VP]),
    "pwn_hip_cloud_export_bound": (C.c_size_t, [_I, _I, _I, _I]),
    "pwn_hip_cloud_export": (_I, [_VP, _VP, _VP, C.c_size_t, C.POINTER(C.c_size_t)]),
    "pwn_hip_cloud_import": (_I, [_VP, _VP, _VP, C.c_size_t]),
    "pwn_hip_depth_u16_to_f32": (_I, [_VP, _VP, _VP, _I, _F]),
    "pwn_hip_depth_f32_to_u16": (_I, [_VP, _VP, _VP, _I, _F]),
    "pwn_hip_depth_scale": (_I, [_VP, _VP, _I, _I, _I, _F, _VP]),
    "pwn_hip_unproject": (_I, [_VP, _VP, _VP, _VP, _I, _I, _VP, _VP]),
    "pwn_hip_project_intervals": (_I, [_VP, _VP, _VP, _I, _I, _VP]),
    "pwn_hip_integral_image": (_I, [_VP, _VP, _VP, _I, _I, _VP]),
    "pwn_hip_convert": (_I, [_VP, _VP, _VP, _I, _I, _VP, _VP, _VP, _I]),
    "pwn_hip_convert_scaled": (_I, [_VP, _VP, _VP, _I, _I, _I, _F, _VP]),
    "pwn_hip_convert_scaled_begin": (_I, [_VP, _VP, _VP, _I, _I, _I, _F, _VP]),
    "pwn_hip_convert_end": (_I, [_VP, _VP]),
    "pwn_hip_convert_export_begin": (_I, [_VP, _VP, _VP, _F, _I, _I, _VP, _VP, C.c_size_t]),
    "pwn_hip_convert_export_end": (_I, [_VP, _VP, C.POINTER(C.c_size_t), C.POINTER(_F)]),
    "pwn_hip_convert_batch": (_I, [_VP, _VP, _VP, _I, _I, _I, _VP]),
    "pwn_hip_convert_batch_u16": (_I, [_VP, _VP, _VP, _F, _I, _I, _I, _VP]),
    "pwn_hip_project": (_I, [_VP, _VP, _VP, _F, _F, _I, _I, _VP, _VP, _VP]),
    "pwn_hip_correspondences": (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, C.POINTER(_I), C.POINTER(_I)]),
    "pwn_hip_linearize": (_I, [_VP, _VP, _VP, _VP, _VP, _I, _VP, _VP, _VP, C.POINTER(_F), C.POINTER(_I)]),
    "pwn_hip_align": (_I, [_VP, _VP, _VP, _VP, _VP]),
    "pwn_hip_align_images": (_I, [_VP, _VP, _VP, _VP, _VP]),
    "pwn_hip_align_batch": (_I, [_VP, _VP, _I, _VP, _VP, _VP, _VP]),
    "pwn_hip_align_with_priors": (_I, [_VP, _VP, _VP, _VP, _I, _VP, _VP]),
    "pwn_hip_align_with_priors_ex": (_I, [_VP, _VP, _VP, _VP, _I, _VP, _VP, _VP]),
    "pwn_hip_align_batch_ex": (_I, [_VP, _VP, _I, _VP, _VP, _VP, _VP, _F, _VP, _VP]),
    "pwn_hip_align_batch_records": (_I, [_VP, _VP, _I, _VP, _VP, _VP, _VP, _I, _VP, _VP]),
    "pwn_hip_convert_align_batch_u16": (_I, [_VP, _VP, _VP, _I, _VP, _VP, _F, _I, _I, _VP, _VP, _VP, _VP, _I, _VP, _VP]),
    "pwn_hip_compute_statistics": (None, [_VP, _VP, _VP, _VP, _VP, _VP]),
    "pwn_hip_match_score": (_I, [_VP, _F, _VP]),
    "pwn_hip_match_batch": (_I, [_VP, _VP, _I, _VP, _VP, _VP, _F, _VP, _VP]),
    "pwn_hip_match_batch_records": (_I, [_VP, _VP, _I, _VP, _VP, _VP, _F, _VP, _I, _VP, _VP, _VP]),
    "pwn_hip_projector_matrices": (None, [_VP, _VP, _VP, _VP, _VP]),
    "pwn_hip_project_point": (_I, [_VP, _VP, _F, _F, _VP, C.POINTER(_I), C.POINTER(_I), C.POINTER(_F)]),
    "pwn_hip_unproject_pixel": (_I, [_VP, _VP, _F, _F, _I, _I, _F, _VP]),
    "pwn_hip_project_interval": (_I, [_VP, _F, _F, _F, _F]),
    "pwn_hip_iso_inverse": (None, [_VP, _VP]),
    "pwn_hip_iso_mul": (None, [_VP, _VP, _VP]),
    "pwn_hip_v2t": (None, [_VP, _VP]),
    "pwn_hip_t2v": (None, [_VP, _VP]),
    "pwn_hip_ldlt_solve6": (None, [_VP, _VP, _VP]),
    "pwn_hip_cloud_gaussians": (_I, [_VP, _VP, _VP, _I, _I, _VP, _F, _F]),
    "pwn_hip_cloud_num_gaussians": (_I, [_VP, _VP, C.POINTER(_I)]),
    "pwn_hip_cloud_download_gaussians": (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "pwn_hip_cloud_add": (_I, [_VP, _VP, _VP, _VP]),
    "pwn_hip_merge": (_I, [_VP, _VP, _VP, _VP, _F, _F, _I, _I, _F, _F, _F, C.POINTER(_I), _VP]),
    "pwn_hip_voxelize": (_I, [_VP, _VP, _F, C.POINTER(_I), _VP]),
    "pwn_hip_cloud_save": (_I, [_VP, _VP, C.c_char_p, _VP, _I, _I]),
    "pwn_hip_cloud_load": (_I, [_VP, _VP, C.c_char_p, _VP]),
    "pwn_hip_last_stage_ms": (_I, [_VP, C.c_char_p, C.POINTER(_F), C.POINTER(_I)]),
    "pwn_hip_set_profiling": (_I, [_VP, _I]),
    "pwn_hip_debug_withhold_carry": (_I, [_VP, _I, _I, _I, _I, _I]),
    "pwn_hip_debug_set_index_shortcut": (_I, [_VP, _I]),
    "pwn_hip_debug_convert_retries": (_I, [_VP, C.POINTER(_I)]),
    "pwn_hip_debug_set_settle_guard": (_I, [_VP, _I]),
    "pwn_hip_debug_projection_fallbacks": (_I, [_VP, C.POINTER(_I)]),
    "pwn_hip_measure_hbm": (_I, [_VP, C.c_size_t, _VP, _VP]),
}

_lib = None


def lib():
    """The loaded C-ABI library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: the HIP extension has not been built "
                "(run `python -m g2o_frontend_amd.build` or `__graft_entry__.build()`); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(L, name)          # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib
