"""Host-side mirror of the reference's pwn_core operator interface over the C-ABI (include/pwn_hip.h).

Class and method names follow g2o_frontend/pwn_core (file:line cited per class) so that code and tests
written against the reference read the same here.  Every compute method is one call into
``libpwn_hip.so``; nothing is computed in Python and nothing falls back to the CPU.

Matrices are numpy arrays indexed [row, col] (converted to the ABI's column-major floats at the
boundary); images are [rows, cols].  Buffers may be numpy arrays (host) or torch CUDA tensors
(device, passed by ``data_ptr()``).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import AlignerParams, AlignResult, AlignStatistics, ConverterParams, MatchResult, Prior, PwnHipError


def _ptr(x):
    if x is None:
        return None
    if hasattr(x, "data_ptr"):                      # torch tensor
        if not x.is_contiguous():
            raise ValueError("tensor must be contiguous")
        return C.c_void_p(x.data_ptr())
    if isinstance(x, np.ndarray):
        if not x.flags["C_CONTIGUOUS"]:
            raise ValueError("array must be C-contiguous")
        return x.ctypes.data_as(C.c_void_p)
    raise TypeError(type(x))


class DeviceBuffer:
    """a caller-owned buffer in HBM; accepted wherever a frame / image pointer is (it has data_ptr(), shape, is_contiguous() like a tensor)"""

    def __init__(self, ctx, array):
        a = np.ascontiguousarray(array)
        self.ctx, self.shape, self.dtype, self.nbytes = ctx, a.shape, a.dtype, a.nbytes
        self._p = C.c_void_p()
        ctx.check(ctx._L.pwn_hip_device_alloc(ctx.h, C.byref(self._p), max(1, a.nbytes)))
        ctx.check(ctx._L.pwn_hip_copy(ctx.h, self._p, a.ctypes.data_as(C.c_void_p), a.nbytes))

    def data_ptr(self): return self._p.value
    def is_contiguous(self): return True

    def frame(self, i):
        """the i-th image of a [n, rows, cols] buffer as a frame argument"""
        return _DeviceView(self, i)

    def copy_from_async(self, array):
        """queue host -> device on the context's copy stream (pwn_hip_copy_async); the next convert call waits for it"""
        assert array.nbytes == self.nbytes and array.flags["C_CONTIGUOUS"]
        self.ctx.check(self.ctx._L.pwn_hip_copy_async(self.ctx.h, self._p, array.ctypes.data_as(C.c_void_p), array.nbytes))

    def numpy(self):
        out = np.empty(self.shape, self.dtype)
        self.ctx.check(self.ctx._L.pwn_hip_copy(self.ctx.h, out.ctypes.data_as(C.c_void_p), self._p, self.nbytes))
        return out

    def free(self):
        if self._p:      # a context that is already closed took its streams along: the buffer is freed without it (plain hipFree)
            self.ctx._L.pwn_hip_device_free(self.ctx.h if self.ctx.h else None, self._p)
        self._p = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class _DeviceView:
    def __init__(self, buf, i):
        self.buf, self.shape = buf, tuple(buf.shape[1:])
        self._off = i * int(np.prod(self.shape)) * buf.dtype.itemsize

    def data_ptr(self): return self.buf.data_ptr() + self._off
    def is_contiguous(self): return True


class _PinnedBlock:
    """owner of one pwn_hip_host_alloc block; freed when the last array view over it is gone (every numpy view reaches it through its
    base chain: view -> memoryview -> the ctypes buffer below, which holds the only reference to this owner)"""

    def __init__(self, nbytes):
        self.ptr = C.c_void_p()
        rc = _lib.lib().pwn_hip_host_alloc(C.byref(self.ptr), nbytes)
        if rc != 0:
            raise _lib.PwnHipError(rc, _lib.lib().pwn_hip_last_error_string(None).decode())

    def __del__(self):
        try:
            if self.ptr:
                _lib.lib().pwn_hip_host_free(self.ptr); self.ptr = None
        except Exception:
            pass


def pinned_empty(shape, dtype=np.float32):
    """numpy array in page-locked host memory (pwn_hip_host_alloc): depth frames handed over from it are copied by asynchronous DMA.
    The memory lives as long as any view of the array does.  Before the last view goes away, make sure no pwn_hip_copy_async from it is
    still in flight (a convert call on that context, or Context.synchronize(), has returned)."""
    dtype = np.dtype(dtype)
    n = int(np.prod(shape))
    nbytes = max(1, n * dtype.itemsize)
    blk = _PinnedBlock(nbytes)
    buf = (C.c_char * nbytes).from_address(blk.ptr.value)
    buf._pwn_owner = blk                                       # the buffer object keeps the block alive, the arrays keep the buffer alive
    return np.frombuffer(buf, dtype=dtype, count=n).reshape(shape)


def pinned_free(a):
    """kept for callers of the earlier interface: the block is released with its last view, nothing to do here"""
    return None


def _nbytes(x) -> int:
    if isinstance(x, DeviceBuffer):
        return int(x.nbytes)
    if hasattr(x, "data_ptr"):
        return int(x.numel() * x.element_size())
    return int(x.nbytes)


def _check_records(records, n, floats, ctx=None):
    """A records buffer the library writes n * floats float32 words into (k_pack_records on the device, or a copy to the host): anything
    smaller, of another type, non-contiguous or on another device would be written out of bounds."""
    if records is None:
        return
    if isinstance(records, DeviceBuffer):
        if records.dtype != np.float32 or records.nbytes < 4 * n * floats:
            raise ValueError(f"records must be a float32 device buffer of at least {n} x {floats} floats")
        return
    if hasattr(records, "data_ptr"):
        import torch
        if records.dtype != torch.float32 or not records.is_contiguous():
            raise ValueError("records must be a contiguous float32 tensor")
        if records.numel() < n * floats:
            raise ValueError(f"records holds {records.numel()} floats, {n} x {floats} are written")
        if records.is_cuda and ctx is not None and records.device.index not in (None, ctx.device):
            raise ValueError(f"records lives on cuda:{records.device.index}, the context on device {ctx.device}")
    elif isinstance(records, np.ndarray):
        if records.dtype != np.float32 or not records.flags["C_CONTIGUOUS"] or not records.flags["WRITEABLE"]:
            raise ValueError("records must be a writable C-contiguous float32 array")
        if records.size < n * floats:
            raise ValueError(f"records holds {records.size} floats, {n} x {floats} are written")
    else:
        raise TypeError(type(records))


def _check_flat(buffer, ctx, writable: bool):
    """A flat-cloud byte buffer handed to pwn_hip_cloud_export / _import (whose section copies are kernels on the context's device): a strided
    view, another element type, a read-only array as destination or a tensor on another GPU would be read or written out of bounds."""
    if isinstance(buffer, DeviceBuffer):
        return
    if hasattr(buffer, "data_ptr"):
        import torch
        if buffer.dtype != torch.uint8 or not buffer.is_contiguous():
            raise ValueError("a flat cloud buffer must be a contiguous uint8 tensor")
        if buffer.is_cuda and buffer.device.index not in (None, ctx.device):
            raise ValueError(f"the flat cloud buffer lives on cuda:{buffer.device.index}, the context on device {ctx.device}")
    elif isinstance(buffer, np.ndarray):
        if buffer.dtype != np.uint8 or not buffer.flags["C_CONTIGUOUS"]:
            raise ValueError("a flat cloud buffer must be a C-contiguous uint8 array")
        if writable and not buffer.flags["WRITEABLE"]:
            raise ValueError("export needs a writable buffer")
    else:
        raise TypeError(type(buffer))


def _colmajor(M, n):
    return np.ascontiguousarray(np.asarray(M, dtype=np.float32).reshape(n, n).T.reshape(-1))


def _set(field, M, n):
    a = _colmajor(M, n)
    C.memmove(field, a.ctypes.data, a.nbytes)            # one copy instead of n*n Python assignments (the tracker sets 3 matrices per frame)


def _from_colmajor(a, n):
    return np.frombuffer(a, dtype=np.float32, count=n * n).reshape(n, n).T.copy()


# numpy view of pwn_hip_align_result (include/pwn_hip.h): zero-copy access to a batch of results
ALIGN_RESULT_DTYPE = np.dtype([("T", np.float32, 16), ("error", np.float32), ("inliers", np.int32), ("iterations", np.int32),
                               ("total_time_ms", np.float32), ("chi2", np.float32, _lib.MAX_ITERATIONS),
                               ("iter_inliers", np.int32, _lib.MAX_ITERATIONS), ("iter_correspondences", np.int32, _lib.MAX_ITERATIONS),
                               ("iter_candidates", np.int32, _lib.MAX_ITERATIONS), ("n_reference", np.int32), ("n_current", np.int32)])
assert ALIGN_RESULT_DTYPE.itemsize == C.sizeof(AlignResult)


OMEGA_STORAGE = {"exact9": 0, "sym6": 1}      # PWN_HIP_OMEGA_EXACT9 / PWN_HIP_OMEGA_SYM6
RECORD_FLOATS = 64                            # PWN_HIP_RECORD_FLOATS
MATCH_RECORD_FLOATS = 72                      # PWN_HIP_MATCH_RECORD_FLOATS


def device_count() -> int:
    return _lib.lib().pwn_hip_device_count()


class Context:
    """One per (GPU, host thread): owns the stream and the device workspaces
    (cf. pwn_cuda createContext, pwn_cuda/cudaaligner.h:59)."""

    DEFAULT_OMEGA_STORAGE = "sym6"            # what pwn_hip_ctx_create sets (include/pwn_hip.h; tests/test_omega_sym6.py holds the two together)

    def __init__(self, device: int = 0, max_rows: int = 480, max_cols: int = 640, max_batch: int = 1, omega_storage: str = None):
        self._L = _lib.lib()
        h = C.c_void_p()
        rc = self._L.pwn_hip_ctx_create(C.byref(h), device, max_rows, max_cols, max_batch)
        if rc:
            raise PwnHipError(rc, self._L.pwn_hip_last_error_string(None).decode())
        self.h = h
        self.device = int(device)
        self.max_rows, self.max_cols, self.max_batch = max_rows, max_cols, max_batch
        self.omega_storage = self.DEFAULT_OMEGA_STORAGE
        if omega_storage is not None and omega_storage != self.omega_storage:
            self.set_omega_storage(omega_storage)

    def check(self, rc):
        if rc:
            raise PwnHipError(rc, self._L.pwn_hip_last_error_string(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            self._L.pwn_hip_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_ptr):
        self.check(self._L.pwn_hip_ctx_set_stream(self.h, C.c_void_p(stream_ptr) if stream_ptr else None))

    def set_subbatch(self, frames, pairs):
        self.check(self._L.pwn_hip_ctx_set_subbatch(self.h, frames, pairs))

    def set_concurrency(self, streams: int):
        self.check(self._L.pwn_hip_ctx_set_concurrency(self.h, streams))

    def set_omega_storage(self, mode: str):
        """storage of the point information matrices of the clouds created from now on: "exact9" (all nine entries as the reference
        evaluates them; every converter output bit-identical) or "sym6" (upper triangle, 24 bytes; include/pwn_hip.h)"""
        if mode not in OMEGA_STORAGE:
            raise ValueError(f"omega_storage must be one of {sorted(OMEGA_STORAGE)}")
        self.check(self._L.pwn_hip_ctx_set_omega_storage(self.h, OMEGA_STORAGE[mode]))
        self.omega_storage = mode

    def upload(self, array):
        """DeviceBuffer holding a copy of a host array (pwn_hip_device_alloc + pwn_hip_copy): a frame resident in HBM without torch"""
        return DeviceBuffer(self, array)

    @staticmethod
    def _stream_ptr(stream):
        if stream is None:
            try:
                import torch
                stream = torch.cuda.current_stream().cuda_stream
            except Exception:
                stream = 0
        return int(getattr(stream, "cuda_stream", stream))

    def signal_stream(self, stream=None):
        """pwn_hip_ctx_signal_stream: what the caller queues on `stream` (default: torch's current stream) from now on runs after everything
        the context has queued so far."""
        sp = self._stream_ptr(stream)
        self.check(self._L.pwn_hip_ctx_signal_stream(self.h, C.c_void_p(sp) if sp else None))

    def set_enqueued_callback(self, fn):
        """pwn_hip_ctx_set_enqueued_callback: fn() runs inside every alignment batch call after its device work is queued and before the call
        waits for it (None switches it off).  An exception raised by fn is kept and re-raised by take_callback_error()."""
        self._cb_error = None
        if fn is None:
            self._cb = None
            self.check(self._L.pwn_hip_ctx_set_enqueued_callback(self.h, None, None))
            return

        def trampoline(_user):
            try:
                fn()
            except BaseException as e:      # must not propagate through the C frames
                self._cb_error = e
        self._cb = C.CFUNCTYPE(None, C.c_void_p)(trampoline)      # kept alive as long as it is installed
        self.check(self._L.pwn_hip_ctx_set_enqueued_callback(self.h, C.cast(self._cb, C.c_void_p), None))

    def take_callback_error(self):
        e, self._cb_error = getattr(self, "_cb_error", None), None
        if e is not None:
            raise e

    def wait_stream(self, stream=None):
        """pwn_hip_ctx_wait_stream: what the context queues from now on runs after the work the caller's stream holds now.  stream: a raw
        hipStream_t value, a torch.cuda.Stream, or None = torch's current stream (the legacy default stream without torch)."""
        if stream is None:
            try:
                import torch
                stream = torch.cuda.current_stream().cuda_stream
            except Exception:
                stream = 0
        elif hasattr(stream, "cuda_stream"):
            stream = stream.cuda_stream
        self.check(self._L.pwn_hip_ctx_wait_stream(self.h, C.c_void_p(int(stream) or None)))

    def synchronize(self):
        self.check(self._L.pwn_hip_ctx_synchronize(self.h))

    def set_profiling(self, on: bool):
        self.check(self._L.pwn_hip_set_profiling(self.h, 1 if on else 0))

    def measure_hbm(self, nbytes: int = 1 << 30):
        """(read GB/s, copy GB/s) of float4 streaming kernels over `nbytes` on this GPU (the copy counts read + written bytes)"""
        r, c = C.c_float(0), C.c_float(0)
        self.check(self._L.pwn_hip_measure_hbm(self.h, nbytes, C.byref(r), C.byref(c)))
        return float(r.value), float(c.value)

    def stage_ms(self, stage: str):
        ms, n = C.c_float(0), C.c_int(0)
        self.check(self._L.pwn_hip_last_stage_ms(self.h, stage.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    # ---- pwn_static.cpp ---------------------------------------------------------------------
    def DepthImage_convert_16UC1_to_32FC1(self, src, scale=0.001):
        """pwn_core/pwn_static.cpp:54-68"""
        src = np.ascontiguousarray(src, dtype=np.uint16)
        dst = np.empty(src.shape, np.float32)
        self.check(self._L.pwn_hip_depth_u16_to_f32(self.h, _ptr(src), _ptr(dst), src.size, scale))
        return dst

    def DepthImage_convert_32FC1_to_16UC1(self, src, scale=1000.0):
        """pwn_core/pwn_static.cpp:38-52"""
        src = np.ascontiguousarray(src, dtype=np.float32)
        dst = np.empty(src.shape, np.uint16)
        self.check(self._L.pwn_hip_depth_f32_to_u16(self.h, _ptr(src), _ptr(dst), src.size, scale))
        return dst

    def DepthImage_scale(self, src, step, maxDepthCov=0.01):
        """pwn_core/pwn_static.cpp:5-36"""
        src = np.ascontiguousarray(src, dtype=np.float32)
        dst = np.empty((src.shape[0] // step, src.shape[1] // step), np.float32)
        self.check(self._L.pwn_hip_depth_scale(self.h, _ptr(src), src.shape[0], src.shape[1], step, maxDepthCov, _ptr(dst)))
        return dst


class Cloud:
    """Device-resident pwn::Cloud (pwn_core/cloud.h:20-187): points, normals, curvature (of Stats) and the
    two information-matrix vectors."""

    def __init__(self, ctx: Context, capacity: int):
        self.ctx = ctx
        h = C.c_void_p()
        ctx.check(ctx._L.pwn_hip_cloud_create(ctx.h, capacity, C.byref(h)))
        self.h = h
        self.capacity = capacity

    def __del__(self):
        try:
            if getattr(self, "h", None):
                # a cloud that outlives its context is freed outright (pwn_hip_cloud_destroy(NULL, h)); ctx_destroy only frees pooled clouds
                self.ctx._L.pwn_hip_cloud_destroy(self.ctx.h if self.ctx.h else None, self.h)
            self.h = None
        except Exception:
            pass

    def size(self) -> int:
        n = C.c_int(0)
        self.ctx.check(self.ctx._L.pwn_hip_cloud_size(self.ctx.h, self.h, C.byref(n)))
        return n.value

    __len__ = size

    def omega_storage(self) -> str:
        """"exact9" or "sym6": how this cloud keeps its point information matrices (Context.set_omega_storage at its creation)"""
        m = C.c_int(0)
        self.ctx.check(self.ctx._L.pwn_hip_cloud_omega_storage(self.ctx.h, self.h, C.byref(m)))
        return "sym6" if m.value == OMEGA_STORAGE["sym6"] else "exact9"

    def upload(self, points, normals, curvature, omega_p, omega_n):
        a = [np.ascontiguousarray(x, dtype=np.float32) for x in (points, normals, curvature, omega_p, omega_n)]
        self.ctx.check(self.ctx._L.pwn_hip_cloud_upload(self.ctx.h, self.h, len(a[0]), *[_ptr(x) for x in a]))

    def arrays(self, stats: bool = False):
        n = self.size()
        out = dict(points=np.empty((n, 4), np.float32), normals=np.empty((n, 4), np.float32),
                   curvature=np.empty(n, np.float32), omega_p=np.empty((n, 16), np.float32),
                   omega_n=np.empty((n, 16), np.float32))
        self.ctx.check(self.ctx._L.pwn_hip_cloud_download(self.ctx.h, self.h, _ptr(out["points"]), _ptr(out["normals"]),
                                                          _ptr(out["curvature"]), _ptr(out["omega_p"]), _ptr(out["omega_n"])))
        if stats:
            out.update(stats=np.empty((n, 16), np.float32), eigenvalues=np.empty((n, 3), np.float32),
                       npoints=np.empty(n, np.int32))
            self.ctx.check(self.ctx._L.pwn_hip_cloud_download_stats(self.ctx.h, self.h, _ptr(out["stats"]),
                                                                    _ptr(out["eigenvalues"]), _ptr(out["npoints"])))
        return out

    # ---- the cloud as one flat buffer (replication to other GPUs: include/pwn_hip.h, pwn_hip_cloud_export) ----
    @staticmethod
    def flatBound(capacity: int, omega_storage: str = "exact9", index_pixels: int = 0, with_omega_n: bool = False) -> int:
        """bytes of a flat buffer that holds any cloud of this capacity"""
        return int(_lib.lib().pwn_hip_cloud_export_bound(int(capacity), OMEGA_STORAGE[omega_storage], int(index_pixels), 1 if with_omega_n else 0))

    def flatSize(self) -> int:
        w = C.c_size_t(0)
        self.ctx.check(self.ctx._L.pwn_hip_cloud_export(self.ctx.h, self.h, None, 0, C.byref(w)))
        return int(w.value)

    def exportFlat(self, buffer) -> int:
        """pwn_hip_cloud_export into `buffer` (uint8 numpy array or CUDA tensor, >= flatSize() bytes); returns the bytes used"""
        _check_flat(buffer, self.ctx, writable=True)
        w = C.c_size_t(0)
        self.ctx.check(self.ctx._L.pwn_hip_cloud_export(self.ctx.h, self.h, _ptr(buffer), _nbytes(buffer), C.byref(w)))
        return int(w.value)

    def importFlat(self, buffer):
        """pwn_hip_cloud_import: this cloud becomes the cloud `buffer` was exported from (same omega storage, capacity >= its points)"""
        _check_flat(buffer, self.ctx, writable=False)
        self.ctx.check(self.ctx._L.pwn_hip_cloud_import(self.ctx.h, self.h, _ptr(buffer), _nbytes(buffer)))

    # the reference's per-field accessors (cloud.h:33-131) as host copies: each is one download of that field
    def _field(self, i, width):
        n = self.size()
        out = [None] * 5
        out[i] = np.empty((n, width) if width > 1 else n, np.float32)
        self.ctx.check(self.ctx._L.pwn_hip_cloud_download(self.ctx.h, self.h, *[_ptr(x) for x in out]))
        return out[i]

    def points(self): return self._field(0, 4)                          # n x 4 (x, y, z, 1)
    def normals(self): return self._field(1, 4)                         # n x 4 (nx, ny, nz, 0)
    def curvatures(self): return self._field(2, 1)                      # Stats::curvature() per point
    def pointInformationMatrix(self): return self._field(3, 16)         # n x 16, column-major 4x4
    def normalInformationMatrix(self): return self._field(4, 16)
    def stats(self): return self.arrays(stats=True)["stats"]            # n x 16 (needs a conversion with keep_stats)
    def traversabilityVector(self): return []                           # cloud.h:99: never written on this path (only pwn_viewer code fills it)

    def transformInPlace(self, T):
        """pwn_core/cloud.cpp:173-186"""
        self.ctx.check(self.ctx._L.pwn_hip_cloud_transform_in_place(self.ctx.h, self.h, _ptr(_colmajor(T, 4))))

    # ---- scene maintenance (pwn_core/cloud.cpp:11-171, gaussian3.h) ----
    def numGaussians(self) -> int:
        n = C.c_int(0)
        self.ctx.check(self.ctx._L.pwn_hip_cloud_num_gaussians(self.ctx.h, self.h, C.byref(n)))
        return n.value

    def gaussians(self):
        """Cloud::gaussians(): mean, cov (column-major 3x3), info_vec, info, flags (1 = moments valid, 2 = information form valid)"""
        n = self.numGaussians()
        out = dict(mean=np.empty((n, 3), np.float32), cov=np.empty((n, 9), np.float32), info_vec=np.empty((n, 3), np.float32),
                   info=np.empty((n, 9), np.float32), flags=np.empty(n, np.int32))
        self.ctx.check(self.ctx._L.pwn_hip_cloud_download_gaussians(self.ctx.h, self.h, _ptr(out["mean"]), _ptr(out["cov"]), _ptr(out["info_vec"]),
                                                                    _ptr(out["info"]), _ptr(out["flags"])))
        return out

    def add(self, cloud: "Cloud", T=None):
        """Cloud::add (cloud.cpp:145-171)"""
        self.ctx.check(self.ctx._L.pwn_hip_cloud_add(self.ctx.h, self.h, cloud.h, _ptr(_colmajor(np.eye(4) if T is None else T, 4))))

    def save(self, filename, T=None, step: int = 1, binary: bool = False) -> bool:
        """Cloud::save (cloud.cpp:84-136)"""
        self.ctx.check(self.ctx._L.pwn_hip_cloud_save(self.ctx.h, self.h, str(filename).encode(), _ptr(_colmajor(np.eye(4) if T is None else T, 4)),
                                                      int(step), 1 if binary else 0))
        return True

    def load(self, filename):
        """Cloud::load (cloud.cpp:25-82) -> the transform stored in the file"""
        T = np.empty(16, np.float32)
        self.ctx.check(self.ctx._L.pwn_hip_cloud_load(self.ctx.h, self.h, str(filename).encode(), _ptr(T)))
        return _from_colmajor(T, 4)


class PinholePointProjector:
    """pwn_core/pinholepointprojector.{h,cpp} + pointprojector.{h,cpp}: parameter holder; project /
    unProject / projectIntervals run on the GPU."""

    def __init__(self, ctx: Context | None = None):
        self.ctx = ctx
        self._K = np.array([[1, 0, 0.5], [0, 1, 0.5], [0, 0, 1]], np.float32)   # pinholepointprojector.cpp:6-9
        self._transform = np.eye(4, dtype=np.float32)
        self._minDistance, self._maxDistance = 0.01, 6.0                        # pointprojector.cpp:9-10
        self._imageRows = self._imageCols = 0
        self._baseline, self._alpha = 0.075, 0.1                                # pinholepointprojector.cpp:10-11 (sensor-noise Gaussians)

    def baseline(self): return self._baseline
    def setBaseline(self, v): self._baseline = float(v)
    def alpha(self): return self._alpha
    def setAlpha(self, v): self._alpha = float(v)

    def cameraMatrix(self): return self._K
    def setCameraMatrix(self, K): self._K = np.asarray(K, np.float32).reshape(3, 3).copy()
    def transform(self): return self._transform
    def setTransform(self, T): self._transform = np.asarray(T, np.float32).reshape(4, 4).copy()
    def minDistance(self): return self._minDistance
    def setMinDistance(self, v): self._minDistance = float(v)
    def maxDistance(self): return self._maxDistance
    def setMaxDistance(self, v): self._maxDistance = float(v)
    def imageRows(self): return self._imageRows
    def imageCols(self): return self._imageCols
    def setImageSize(self, rows, cols): self._imageRows, self._imageCols = int(rows), int(cols)

    def scale(self, s):
        """pinholepointprojector.cpp:149-154"""
        self._K[:2, :] = self._K[:2, :] * np.float32(s)
        self._imageRows = int(np.float32(self._imageRows) * np.float32(s))
        self._imageCols = int(np.float32(self._imageCols) * np.float32(s))

    def inverseCameraMatrix(self): return self.matrices()[2]                                     # pinholepointprojector.h:61

    def matrices(self):
        """_updateMatrices (pinholepointprojector.cpp:17-31): (KRt, iKRt, iK)"""
        KRt = np.empty(16, np.float32); iKRt = np.empty(16, np.float32); iK = np.empty(9, np.float32)
        _lib.lib().pwn_hip_projector_matrices(_ptr(_colmajor(self._K, 3)), _ptr(_colmajor(self._transform, 4)), _ptr(KRt), _ptr(iKRt), _ptr(iK))
        return KRt.reshape(4, 4).T.copy(), iKRt.reshape(4, 4).T.copy(), iK.reshape(3, 3).T.copy()

    # the single-point forms (pinholepointprojector.h:174,187,200): host code, the kernels' own expressions
    def projectPoint(self, p):
        """project(x, y, f, p) -> (valid, x, y, depth); the image bounds are the caller's test, as in the reference"""
        x, y, d = C.c_int(0), C.c_int(0), C.c_float(0)
        pt = np.ascontiguousarray(np.asarray(p, np.float32).reshape(-1)[:3])
        ok = _lib.lib().pwn_hip_project_point(_ptr(_colmajor(self._K, 3)), _ptr(_colmajor(self._transform, 4)), self._minDistance, self._maxDistance,
                                              _ptr(pt), C.byref(x), C.byref(y), C.byref(d))
        return bool(ok), x.value, y.value, d.value

    def unProjectPixel(self, x, y, d):
        """unProject(p, x, y, d) -> (valid, point[3]); x = column, y = row"""
        out = np.zeros(3, np.float32)
        ok = _lib.lib().pwn_hip_unproject_pixel(_ptr(_colmajor(self._K, 3)), _ptr(_colmajor(self._transform, 4)), self._minDistance, self._maxDistance,
                                                int(x), int(y), float(d), _ptr(out))
        return bool(ok), out

    def projectInterval(self, x, y, d, worldRadius):
        """projectInterval(x, y, d, worldRadius) -> pixels, -1 for a depth outside [min, max]"""
        return int(_lib.lib().pwn_hip_project_interval(_ptr(_colmajor(self._K, 3)), self._minDistance, self._maxDistance, float(d), float(worldRadius)))

    def project(self, cloud: Cloud):
        """project(indexImage, depthImage, points) (pinholepointprojector.cpp:33-66) -> (index, depth)"""
        ctx = cloud.ctx
        idx = np.empty((self._imageRows, self._imageCols), np.int32)
        dep = np.empty((self._imageRows, self._imageCols), np.float32)
        ctx.check(ctx._L.pwn_hip_project(ctx.h, _ptr(_colmajor(self._K, 3)), _ptr(_colmajor(self._transform, 4)), self._minDistance,
                                         self._maxDistance, self._imageRows, self._imageCols, cloud.h, _ptr(idx), _ptr(dep)))
        return idx, dep

    def _params(self, world_radius=0.1) -> ConverterParams:
        p = ConverterParams()
        _lib.lib().pwn_hip_default_converter_params(C.byref(p))
        _set(p.K, self._K, 3)
        p.min_distance, p.max_distance, p.world_radius = self._minDistance, self._maxDistance, world_radius
        return p

    def unProject(self, cloud: Cloud, depthImage):
        """unProject(points, gaussians, indexImage, depthImage) (pinholepointprojector.cpp:93-133) -> index image"""
        ctx = cloud.ctx
        depth = depthImage if hasattr(depthImage, "data_ptr") else np.ascontiguousarray(depthImage, np.float32)
        rows, cols = depth.shape
        idx = np.empty((rows, cols), np.int32)
        p = self._params()
        ctx.check(ctx._L.pwn_hip_unproject(ctx.h, C.byref(p), _ptr(_colmajor(self._transform, 4)), _ptr(depth), rows, cols, cloud.h, _ptr(idx)))
        return idx

    def projectIntervals(self, ctx: Context, depthImage, worldRadius):
        """projectIntervals (pinholepointprojector.cpp:135-147) -> interval image"""
        depth = np.ascontiguousarray(depthImage, np.float32)
        rows, cols = depth.shape
        out = np.empty((rows, cols), np.int32)
        p = self._params(worldRadius)
        ctx.check(ctx._L.pwn_hip_project_intervals(ctx.h, C.byref(p), _ptr(depth), rows, cols, _ptr(out)))
        return out


class StatsCalculatorIntegralImage:
    """pwn_core/statscalculatorintegralimage.{h,cpp}: parameters (defaults :6-12)."""

    def __init__(self):
        self._worldRadius, self._maxImageRadius, self._minImageRadius = 0.1, 30, 10
        self._minPoints, self._curvatureThreshold = 50, 0.02

    def setWorldRadius(self, v): self._worldRadius = float(v)
    def worldRadius(self): return self._worldRadius
    def setMaxImageRadius(self, v): self._maxImageRadius = int(v)
    def setMinImageRadius(self, v): self._minImageRadius = int(v)
    def setMinPoints(self, v): self._minPoints = int(v)
    def setCurvatureThreshold(self, v): self._curvatureThreshold = float(v)
    def maxImageRadius(self): return self._maxImageRadius
    def minImageRadius(self): return self._minImageRadius
    def minPoints(self): return self._minPoints
    def curvatureThreshold(self): return self._curvatureThreshold

    @staticmethod
    def integralImage(cloud: Cloud, indexImage):
        """PointIntegralImage::compute (pointintegralimage.cpp:7-44): 10 planes [10, rows, cols]"""
        ctx = cloud.ctx
        idx = np.ascontiguousarray(indexImage, np.int32)
        out = np.empty((10,) + idx.shape, np.float32)
        ctx.check(ctx._L.pwn_hip_integral_image(ctx.h, _ptr(idx), cloud.h, idx.shape[0], idx.shape[1], _ptr(out)))
        return out


class _InformationMatrixCalculator:
    def __init__(self, flat, nonflat, thr):
        self._flat, self._nonflat, self._curvatureThreshold = list(flat), list(nonflat), thr

    def setCurvatureThreshold(self, v): self._curvatureThreshold = float(v)
    def curvatureThreshold(self): return self._curvatureThreshold
    def setFlatInformationMatrix(self, diag): self._flat = [float(x) for x in diag]
    def setNonFlatInformationMatrix(self, diag): self._nonflat = [float(x) for x in diag]
    def flatInformationMatrix(self): return np.diag(np.asarray(self._flat, np.float32))          # informationmatrixcalculator.h:58,74: diagonal matrices
    def nonFlatInformationMatrix(self): return np.diag(np.asarray(self._nonflat, np.float32))


class PointInformationMatrixCalculator(_InformationMatrixCalculator):
    """pwn_core/informationmatrixcalculator.h:95-118 (defaults :106-110)"""

    def __init__(self): super().__init__((1000.0, 1.0, 1.0), (1.0, 1.0, 1.0), 0.02)


class NormalInformationMatrixCalculator(_InformationMatrixCalculator):
    """pwn_core/informationmatrixcalculator.h:130-153 (defaults :141-145)"""

    def __init__(self): super().__init__((100.0, 100.0, 100.0), (1.0, 1.0, 1.0), 0.02)


class DepthImageConverterIntegralImage:
    """pwn_core/depthimageconverterintegralimage.{h,cpp}: compute(cloud, depthImage, sensorOffset)."""

    def __init__(self, projector, statsCalculator, pointInformationMatrixCalculator, normalInformationMatrixCalculator):
        self._projector, self._stats = projector, statsCalculator
        self._pinfo, self._ninfo = pointInformationMatrixCalculator, normalInformationMatrixCalculator
        self._indexImage = None
        self._intervalImage = None

    def projector(self): return self._projector
    def setProjector(self, p): self._projector = p
    def statsCalculator(self): return self._stats                                                # depthimageconverter.h:63-104: the collaborators
    def setStatsCalculator(self, s): self._stats = s
    def pointInformationMatrixCalculator(self): return self._pinfo
    def setPointInformationMatrixCalculator(self, c): self._pinfo = c
    def normalInformationMatrixCalculator(self): return self._ninfo
    def setNormalInformationMatrixCalculator(self, c): self._ninfo = c
    def indexImage(self): return self._indexImage
    def intervalImage(self): return self._intervalImage

    def params(self, sensorOffset=None) -> ConverterParams:
        p = self._projector._params(self._stats._worldRadius)
        p.min_image_radius, p.max_image_radius = self._stats._minImageRadius, self._stats._maxImageRadius
        p.min_points, p.stats_curvature_threshold = self._stats._minPoints, self._stats._curvatureThreshold
        p.point_info_curvature_threshold = self._pinfo._curvatureThreshold
        p.normal_info_curvature_threshold = self._ninfo._curvatureThreshold
        for i in range(3):
            p.point_flat_diag[i], p.point_nonflat_diag[i] = self._pinfo._flat[i], self._pinfo._nonflat[i]
            p.normal_flat_diag[i], p.normal_nonflat_diag[i] = self._ninfo._flat[i], self._ninfo._nonflat[i]
        _set(p.sensor_offset, np.eye(4) if sensorOffset is None else sensorOffset, 4)
        return p

    def compute(self, cloud: Cloud, depthImage, sensorOffset=None, keep_stats: bool = False, images: bool = True, gaussians: bool = False):
        ctx = cloud.ctx
        depth = depthImage if hasattr(depthImage, "data_ptr") else np.ascontiguousarray(depthImage, np.float32)
        rows, cols = depth.shape
        # side effects of the reference: projector image size set, transform reset (depthimageconverterintegralimage.cpp:30,38)
        self._projector.setImageSize(rows, cols)
        self._projector.setTransform(np.eye(4, dtype=np.float32))
        p = self.params(sensorOffset)
        idx = np.empty((rows, cols), np.int32) if images else None
        itv = np.empty((rows, cols), np.int32) if images else None
        ctx.check(ctx._L.pwn_hip_convert(ctx.h, C.byref(p), _ptr(depth), rows, cols, cloud.h, _ptr(idx), _ptr(itv), 1 if keep_stats else 0))
        self._indexImage, self._intervalImage = idx, itv
        if gaussians:      # the Gaussian half of unProject (pinholepointprojector.cpp:104-123); the reference always computes it
            ctx.check(ctx._L.pwn_hip_cloud_gaussians(ctx.h, C.byref(p), _ptr(depth), rows, cols, cloud.h, self._projector._baseline, self._projector._alpha))

    def computeExportBegin(self, cloud: Cloud, rawFrame, raw_scale=0.001, flat=None, sensorOffset=None):
        """The look-ahead of a sharded PwnCloser::processPartition (pwn_hip_convert_export_begin): returns at once; the library's helper thread
        converts the uint16 frame into `cloud` and then writes the cloud's flat form into `flat` (uint8 CUDA tensor / numpy array, or None)
        while the caller goes on using the context.  computeExportEnd(ticket) -> (bytes written, job milliseconds)."""
        ctx = cloud.ctx
        rows, cols = rawFrame.shape
        p = self.params(sensorOffset)
        if flat is not None:
            _check_flat(flat, ctx, writable=True)
        ctx.check(ctx._L.pwn_hip_convert_export_begin(ctx.h, C.byref(p), _ptr(rawFrame), raw_scale, rows, cols, cloud.h, _ptr(flat), _nbytes(flat) if flat is not None else 0))
        return dict(ctx=ctx, cloud=cloud, frame=rawFrame, flat=flat)

    @staticmethod
    def computeExportEnd(ticket):
        ctx, cloud = ticket["ctx"], ticket["cloud"]
        w, ms = C.c_size_t(0), C.c_float(0)
        ctx.check(ctx._L.pwn_hip_convert_export_end(ctx.h, cloud.h, C.byref(w), C.byref(ms)))
        ticket["frame"] = None
        return int(w.value), float(ms.value)

    @staticmethod
    def batchHandles(clouds, depthImages):
        """(cloud handle array, frame pointer array) for computeBatch(..., prepared=...): build once, reuse every call."""
        n = len(clouds)
        return (C.c_void_p * n)(*[c.h for c in clouds]), (C.c_void_p * n)(*[_ptr(d) for d in depthImages]), n, tuple(depthImages[0].shape)

    def computeBatch(self, clouds, depthImages, sensorOffset=None, raw_scale=None, prepared=None):
        """n independent frames in one call.  depthImages: list of float32 [rows, cols] arrays / CUDA tensors,
        or uint16 millimetre frames when raw_scale is given (fuses DepthImage_convert_16UC1_to_32FC1)."""
        ctx = clouds[0].ctx
        p = self.params(sensorOffset)
        if prepared is not None:
            handles, ptrs, n, (rows, cols) = prepared
        else:
            handles, ptrs, n, (rows, cols) = self.batchHandles(clouds, depthImages)
        if raw_scale is None:
            ctx.check(ctx._L.pwn_hip_convert_batch(ctx.h, C.byref(p), ptrs, n, rows, cols, handles))
        else:
            ctx.check(ctx._L.pwn_hip_convert_batch_u16(ctx.h, C.byref(p), ptrs, raw_scale, n, rows, cols, handles))


class Merger:
    """pwn_core/merger.{h,cpp}: merge(cloud, transform) fuses the points of a scene cloud that fall on the same pixel of a virtual
    view (depth and normal compatible) through their sensor-noise Gaussians and drops the fused ones."""

    def __init__(self):
        self._distanceThreshold = 0.1                                          # merger.cpp:6-8
        self._normalThreshold = float(np.cos(np.float32(10 * np.pi / 180.0)))
        self._maxPointDepth = 10.0
        self._depthImageConverter = None
        self._rows = self._cols = 0
        self._collapsedIndices = None

    def distanceThreshold(self): return self._distanceThreshold
    def setDistanceThreshold(self, v): self._distanceThreshold = float(v)
    def normalThreshold(self): return self._normalThreshold
    def setNormalThreshold(self, v): self._normalThreshold = float(v)
    def maxPointDepth(self): return self._maxPointDepth
    def setMaxPointDepth(self, v): self._maxPointDepth = float(v)
    def depthImageConverter(self): return self._depthImageConverter
    def setDepthImageConverter(self, c): self._depthImageConverter = c
    def imageSize(self): return (self._rows, self._cols)
    def setImageSize(self, r, c): self._rows, self._cols = int(r), int(c)
    def collapsedIndices(self): return self._collapsedIndices

    def merge(self, cloud: Cloud, transform=None):
        assert self._rows > 0 and self._cols > 0, "Merger: _indexImage has zero size"
        assert self._depthImageConverter is not None, "Merger: missing _depthImageConverter"
        proj = self._depthImageConverter.projector()
        T = np.eye(4, dtype=np.float32) if transform is None else np.asarray(transform, np.float32)
        proj.setTransform(T)                                                    # merger.cpp:20-21 (side effect on the shared projector)
        ctx = cloud.ctx
        n = cloud.size()
        collapsed = np.empty(max(n, 1), np.int32)
        k = C.c_int(0)
        ctx.check(ctx._L.pwn_hip_merge(ctx.h, cloud.h, _ptr(_colmajor(proj.cameraMatrix(), 3)), _ptr(_colmajor(T, 4)), proj.minDistance(),
                                       proj.maxDistance(), self._rows, self._cols, self._distanceThreshold, self._normalThreshold,
                                       self._maxPointDepth, C.byref(k), _ptr(collapsed)))
        self._collapsedIndices = collapsed[:n]
        return k.value


class VoxelCalculator:
    """pwn_core/voxelcalculator.{h,cpp}: keeps the first point of every voxel of side `resolution`."""

    def __init__(self):
        self._resolution = 0.01                                                 # voxelcalculator.h:53
        self._kept = None

    def resolution(self): return self._resolution
    def setResolution(self, v): self._resolution = float(v)
    def keptIndices(self): return self._kept

    def compute(self, cloud: Cloud, resolution=None):
        res = self._resolution if resolution is None else float(resolution)
        ctx = cloud.ctx
        kept = np.empty(max(cloud.size(), 1), np.int32)
        k = C.c_int(0)
        ctx.check(ctx._L.pwn_hip_voxelize(ctx.h, cloud.h, res, C.byref(k), _ptr(kept)))
        self._kept = kept[:k.value].copy()
        return k.value


class CorrespondenceFinder:
    """pwn_core/correspondencefinder.{h,cpp}: parameters (defaults .cpp:9-18) + compute()."""

    def __init__(self):
        self._inlierDistanceThreshold = 0.5
        self._inlierNormalAngularThreshold = float(np.float32(np.cos(np.pi / 6)))
        self._flatCurvatureThreshold, self._inlierCurvatureRatioThreshold = 0.02, 1.3
        self._rows = self._cols = 0
        self._correspondences = None
        self._numCorrespondences = 0
        self._images = {}

    def setInlierDistanceThreshold(self, v): self._inlierDistanceThreshold = float(v)
    def setInlierNormalAngularThreshold(self, v): self._inlierNormalAngularThreshold = float(v)
    def setFlatCurvatureThreshold(self, v): self._flatCurvatureThreshold = float(v)
    def setInlierCurvatureRatioThreshold(self, v): self._inlierCurvatureRatioThreshold = float(v)
    def inlierDistanceThreshold(self): return self._inlierDistanceThreshold
    def squaredThreshold(self): return float(np.float32(self._inlierDistanceThreshold) * np.float32(self._inlierDistanceThreshold))      # correspondencefinder.h:133
    def inlierNormalAngularThreshold(self): return self._inlierNormalAngularThreshold
    def flatCurvatureThreshold(self): return self._flatCurvatureThreshold
    def inlierCurvatureRatioThreshold(self): return self._inlierCurvatureRatioThreshold
    def setImageSize(self, rows, cols): self._rows, self._cols = int(rows), int(cols)
    def imageRows(self): return self._rows
    def imageCols(self): return self._cols
    def numCorrespondences(self): return self._numCorrespondences
    def correspondences(self): return self._correspondences
    def referenceIndexImage(self): return self._images.get("ref_index")
    def currentIndexImage(self): return self._images.get("cur_index")
    def referenceDepthImage(self): return self._images.get("ref_depth")
    def currentDepthImage(self): return self._images.get("cur_depth")


class Linearizer:
    """pwn_core/linearizer.{h,cpp}: parameters (defaults .cpp:9-15); H/b/error/inliers of the last update."""

    def __init__(self):
        self._inlierMaxChi2, self._robustKernel = 9e3, True
        self._H = np.zeros((6, 6), np.float32); self._b = np.zeros(6, np.float32)
        self._error, self._inliers = 0.0, 0
        self._T = np.eye(4, dtype=np.float32)
        self._aligner = None

    def setAligner(self, a): self._aligner = a
    def aligner(self): return self._aligner
    def inlierMaxChi2(self): return self._inlierMaxChi2
    def robustKernel(self): return self._robustKernel
    def setInlierMaxChi2(self, v): self._inlierMaxChi2 = float(v)
    def setRobustKernel(self, v): self._robustKernel = bool(v)
    def setT(self, T):
        self._T = np.asarray(T, np.float32).reshape(4, 4).copy(); self._T[3] = (0, 0, 0, 1)   # linearizer.h:62-65
    def T(self): return self._T                                                                  # linearizer.h:55
    def H(self): return self._H
    def b(self): return self._b
    def error(self): return self._error
    def inliers(self): return self._inliers


class Aligner:
    """pwn_core/aligner.{h,cpp}.  align() runs the whole Gauss-Newton loop on the GPU (pwn_hip_align)."""

    def __init__(self, ctx: Context):
        self.ctx = ctx
        self._projector = self._linearizer = self._correspondenceFinder = None
        self._referenceCloud = self._currentCloud = None
        self._outerIterations, self._innerIterations = 10, 1          # aligner.cpp:19-20
        I = np.eye(4, dtype=np.float32)
        self._T, self._initialGuess = I.copy(), I.copy()
        self._referenceSensorOffset, self._currentSensorOffset = I.copy(), I.copy()
        self._totalTime, self._error, self._inliers = 0.0, 0.0, 0
        self._result = None
        self._priors = []
        self._omega = np.zeros((6, 6), np.float32); self._mean = np.zeros(6, np.float32); self._statistics = None
        self._translationalEigenRatio = self._rotationalEigenRatio = float(np.finfo(np.float32).max)
        self._rotationalMinEigenRatio = self._translationalMinEigenRatio = 50.0       # aligner.cpp:29-30

    @staticmethod
    def _iso(T):
        T = np.asarray(T, np.float32).reshape(4, 4).copy(); T[3] = (0, 0, 0, 1); return T     # aligner.h:130-190 force the last row

    def setProjector(self, p): self._projector = p
    def setLinearizer(self, l): self._linearizer = l; l.setAligner(self)
    def setCorrespondenceFinder(self, f): self._correspondenceFinder = f
    def projector(self): return self._projector
    def linearizer(self): return self._linearizer
    def correspondenceFinder(self): return self._correspondenceFinder
    def setReferenceCloud(self, c): self._referenceCloud = c; self.clearPriors()     # aligner.h:60-63 (setting a cloud clears the priors)
    def setCurrentCloud(self, c): self._currentCloud = c; self.clearPriors()         # aligner.h:77-80
    def referenceCloud(self): return self._referenceCloud
    def currentCloud(self): return self._currentCloud
    def setOuterIterations(self, n): self._outerIterations = int(n)
    def setInnerIterations(self, n): self._innerIterations = int(n)
    def outerIterations(self): return self._outerIterations
    def innerIterations(self): return self._innerIterations
    def initialGuess(self): return self._initialGuess
    def sensorOffset(self): return self._referenceSensorOffset                                   # aligner.h:141: the reference sensor offset
    def referenceSensorOffset(self): return self._referenceSensorOffset
    def currentSensorOffset(self): return self._currentSensorOffset
    def translationalMinEigenRatio(self): return self._translationalMinEigenRatio
    def rotationalMinEigenRatio(self): return self._rotationalMinEigenRatio
    # aligner.h:216-239: _debug only switches the reference's terminal output on, _minInliers is set by the constructor and never read (aligner.cpp:28): both
    # are kept as inert state so that configuration code written for the reference runs unchanged
    def debug(self): return getattr(self, "_debug", False)
    def setDebug(self, v): self._debug = bool(v)
    def minInliers(self): return getattr(self, "_minInliers", 100)
    def setMinInliers(self, v): self._minInliers = int(v)
    def setInitialGuess(self, T): self._initialGuess = self._iso(T)
    def setSensorOffset(self, T): self._referenceSensorOffset = self._iso(T); self._currentSensorOffset = self._iso(T)
    def setReferenceSensorOffset(self, T): self._referenceSensorOffset = self._iso(T)
    def setCurrentSensorOffset(self, T): self._currentSensorOffset = self._iso(T)
    def T(self): return self._T
    def error(self): return self._error
    def inliers(self): return self._inliers
    def totalTime(self): return self._totalTime
    def result(self): return self._result

    def params(self) -> AlignerParams:
        assert self._projector is not None and self._linearizer is not None and self._correspondenceFinder is not None, "Aligner: missing collaborator"
        p = AlignerParams()
        _lib.lib().pwn_hip_default_aligner_params(C.byref(p))
        pr, f, l = self._projector, self._correspondenceFinder, self._linearizer
        _set(p.K, pr._K, 3)
        p.min_distance, p.max_distance = pr._minDistance, pr._maxDistance
        p.rows, p.cols = pr._imageRows, pr._imageCols
        p.inlier_distance_threshold = f._inlierDistanceThreshold
        p.inlier_normal_angular_threshold = f._inlierNormalAngularThreshold
        p.flat_curvature_threshold = f._flatCurvatureThreshold
        p.inlier_curvature_ratio_threshold = f._inlierCurvatureRatioThreshold
        p.inlier_max_chi2 = l._inlierMaxChi2
        p.robust_kernel = 1 if l._robustKernel else 0
        p.outer_iterations, p.inner_iterations = self._outerIterations, self._innerIterations
        _set(p.reference_sensor_offset, self._referenceSensorOffset, 4)
        _set(p.current_sensor_offset, self._currentSensorOffset, 4)
        _set(p.initial_guess, self._initialGuess, 4)
        return p

    @staticmethod
    def _unpack(r: AlignResult):
        n = r.iterations
        return dict(T=_from_colmajor(r.T, 4), error=r.error, inliers=r.inliers, iterations=n, total_time_ms=r.total_time_ms,
                    chi2=np.frombuffer(r.chi2, np.float32, n).copy(), iter_inliers=np.frombuffer(r.iter_inliers, np.int32, n).copy(),
                    C=np.frombuffer(r.iter_correspondences, np.int32, n).copy(), K=np.frombuffer(r.iter_candidates, np.int32, n).copy(),
                    n_reference=r.n_reference, n_current=r.n_current)

    def addRelativePrior(self, mean, informationMatrix):
        """aligner.cpp:34-36"""
        self._priors.append((0, np.asarray(mean, np.float32).copy(), np.eye(4, dtype=np.float32), np.asarray(informationMatrix, np.float32).copy()))

    def addAbsolutePrior(self, referenceTransform, mean, informationMatrix):
        """aligner.cpp:38-40"""
        self._priors.append((1, np.asarray(mean, np.float32).copy(), np.asarray(referenceTransform, np.float32).copy(), np.asarray(informationMatrix, np.float32).copy()))

    def clearPriors(self): self._priors = []                                    # aligner.cpp:42-47

    def omega(self): return self._omega                                        # aligner.h:314
    def translationalEigenRatio(self): return self._translationalEigenRatio
    def rotationalEigenRatio(self): return self._rotationalEigenRatio
    def setTranslationalMinEigenRatio(self, v): self._translationalMinEigenRatio = float(v)
    def setRotationalMinEigenRatio(self, v): self._rotationalMinEigenRatio = float(v)

    def solutionValid(self) -> bool:
        """the eigen-ratio test of aligner.cpp:128-129 (needs align(statistics=True))"""
        return not (self._rotationalEigenRatio > self._rotationalMinEigenRatio or self._translationalEigenRatio > self._translationalMinEigenRatio)

    def align(self, images: bool = False, statistics: bool = False):
        """aligner.cpp:49-150; statistics=True also runs _computeStatistics (:127,152-199)"""
        assert self._referenceCloud is not None and self._currentCloud is not None, "Aligner: missing cloud"
        p = self.params()
        r = AlignResult()
        q = AlignStatistics() if statistics else None
        if self._priors:
            # priors change the normal equations of every iteration (aligner.cpp:96-108); _computeStatistics still runs afterwards (:127)
            arr = (Prior * len(self._priors))()
            for a, (kind, mean, reft, info) in zip(arr, self._priors):
                a.kind = kind; _set(a.mean, mean, 4); _set(a.reference_transform, reft, 4); _set(a.information, info, 6)
            self.ctx.check(self.ctx._L.pwn_hip_align_with_priors_ex(self.ctx.h, C.byref(p), self._referenceCloud.h, self._currentCloud.h, len(arr), arr,
                                                                    C.byref(r), C.byref(q) if statistics else None))
        elif statistics:
            refs = (C.c_void_p * 1)(self._referenceCloud.h); curs = (C.c_void_p * 1)(self._currentCloud.h)
            self.ctx.check(self.ctx._L.pwn_hip_align_batch_ex(self.ctx.h, C.byref(p), 1, refs, curs, None, C.byref(r), 0.0, None, C.byref(q)))
        else:
            self.ctx.check(self.ctx._L.pwn_hip_align(self.ctx.h, C.byref(p), self._referenceCloud.h, self._currentCloud.h, C.byref(r)))
        if statistics:
            self._omega = _from_colmajor(q.omega, 6); self._mean = np.array(list(q.mean), np.float32)
            self._translationalEigenRatio, self._rotationalEigenRatio = q.translational_eigen_ratio, q.rotational_eigen_ratio
            self._statistics = dict(mean=self._mean, omega=self._omega, translationalEigenRatio=q.translational_eigen_ratio,
                                    rotationalEigenRatio=q.rotational_eigen_ratio, H=_from_colmajor(q.H, 6), b=np.array(list(q.b), np.float32),
                                    error=q.error, inliers=q.inliers)
            self._linearizer._H, self._linearizer._b = self._statistics["H"], self._statistics["b"]
        self._result = self._unpack(r)
        self._T, self._error, self._inliers = self._result["T"], r.error, r.inliers
        self._totalTime = r.total_time_ms
        self._linearizer._error, self._linearizer._inliers = r.error, r.inliers
        # Aligner::align hands the linearizer the inverse transform of every iteration and, in _computeStatistics, of the result
        # (aligner.cpp:90,165-167): after align(), linearizer()->T() is the final T^-1 with its last row forced
        self._linearizer.setT(iso_inverse(self._T))
        if images:
            f = self._correspondenceFinder
            ri = np.empty((p.rows, p.cols), np.int32); ci = np.empty((p.rows, p.cols), np.int32)
            rd = np.empty((p.rows, p.cols), np.float32); cd = np.empty((p.rows, p.cols), np.float32)
            self.ctx.check(self.ctx._L.pwn_hip_align_images(self.ctx.h, _ptr(ri), _ptr(rd), _ptr(ci), _ptr(cd)))
            f._images = dict(ref_index=ri, ref_depth=rd, cur_index=ci, cur_depth=cd)
        return self._result

    def alignBatch(self, references, currents, initialGuesses=None, raw=False, prepared=None):
        """n independent alignments with this aligner's parameters (the loop-closure candidate batch,
        pwn_tracker/pwn_closer.cpp:92-111).  raw=True returns the results as one numpy structured array
        (ALIGN_RESULT_DTYPE, T column-major) instead of a list of dicts; prepared = (refs, curs, n) handle arrays."""
        p = self.params()
        if prepared is not None:
            refs, curs, n = prepared
        else:
            n = len(references)
            refs = (C.c_void_p * n)(*[c.h for c in references])
            curs = (C.c_void_p * n)(*[c.h for c in currents])
        res = (AlignResult * n)()
        g = None
        if initialGuesses is not None:
            g = np.ascontiguousarray(np.stack([_colmajor(self._iso(T), 4) for T in initialGuesses]), np.float32)
        self.ctx.check(self.ctx._L.pwn_hip_align_batch(self.ctx.h, C.byref(p), n, refs, curs, _ptr(g), res))
        if raw:
            return np.frombuffer(res, dtype=ALIGN_RESULT_DTYPE, count=n)
        return [self._unpack(r) for r in res]

    def alignBatchRecords(self, references, currents, records, pair_ids=None, first_pair_id=0, initialGuesses=None, want_results=True, prepared=None):
        """alignBatch whose results also (or only) leave as fixed-size records written on the device (include/pwn_hip.h: PWN_HIP_RECORD_FLOATS):
        records = a float32 [n, 64] CUDA tensor (e.g. what an all-gather sends) or numpy array; returns the structured result array or None."""
        p = self.params()
        refs, curs, n = prepared if prepared is not None else ((C.c_void_p * len(references))(*[c.h for c in references]),
                                                               (C.c_void_p * len(currents))(*[c.h for c in currents]), len(references))
        res = (AlignResult * n)() if want_results else None
        g = None
        if initialGuesses is not None:
            g = np.ascontiguousarray(np.stack([_colmajor(self._iso(T), 4) for T in initialGuesses]), np.float32)
        ids = None if pair_ids is None else np.ascontiguousarray(pair_ids, np.int32)
        _check_records(records, n, RECORD_FLOATS, self.ctx)
        if ids is not None and ids.size < n:
            raise ValueError("pair_ids shorter than the batch")
        self.ctx.check(self.ctx._L.pwn_hip_align_batch_records(self.ctx.h, C.byref(p), n, refs, curs, _ptr(g), _ptr(ids), int(first_pair_id), res, _ptr(records)))
        return np.frombuffer(res, dtype=ALIGN_RESULT_DTYPE, count=n) if want_results else None

    def convertAlignBatch(self, converter, references, currents, refFrames, curFrames, raw_scale=0.001, records=None, pair_ids=None, first_pair_id=0,
                          initialGuesses=None, want_results=True, prepared=None):
        """One candidate batch from raw uint16 frames as one submission (pwn_hip_convert_align_batch_u16): per pair makeCloud of both frames, then
        align; sub-batch k converts while sub-batch k-1 aligns.  prepared = (refs, curs, ref frame ptrs, cur frame ptrs, n, (rows, cols))."""
        if prepared is None:
            prepared = self.convertAlignHandles(references, currents, refFrames, curFrames)
        refs, curs, rf, cf, n, (rows, cols) = prepared[:6]
        # the two parameter structs cost ~35 us of Python per call: a caller that repeats a call with unchanged objects passes them along
        cp, p = prepared[6] if len(prepared) > 6 else (converter.params(None), self.params())
        res = (AlignResult * n)() if want_results else None
        g = None
        if initialGuesses is not None:
            g = np.ascontiguousarray(np.stack([_colmajor(self._iso(T), 4) for T in initialGuesses]), np.float32)
        ids = None if pair_ids is None else np.ascontiguousarray(pair_ids, np.int32)
        _check_records(records, n, RECORD_FLOATS, self.ctx)
        if ids is not None and ids.size < n:
            raise ValueError("pair_ids shorter than the batch")
        self.ctx.check(self.ctx._L.pwn_hip_convert_align_batch_u16(self.ctx.h, C.byref(cp), C.byref(p), n, rf, cf, raw_scale, rows, cols, refs, curs, _ptr(g),
                                                                   _ptr(ids), int(first_pair_id), res, _ptr(records)))
        return np.frombuffer(res, dtype=ALIGN_RESULT_DTYPE, count=n) if want_results else None

    def convertAlignHandles(self, references, currents, refFrames, curFrames, converter=None):
        """handle / pointer arrays for convertAlignBatch(..., prepared=...); with `converter`, the parameter structs of both objects as they
        are NOW ride along (rebuild after changing a parameter)"""
        n = len(references)
        out = ((C.c_void_p * n)(*[c.h for c in references]), (C.c_void_p * n)(*[c.h for c in currents]),
               (C.c_void_p * n)(*[_ptr(d) for d in refFrames]), (C.c_void_p * n)(*[_ptr(d) for d in curFrames]), n, tuple(refFrames[0].shape))
        return out + ((converter.params(None), self.params()),) if converter is not None else out

    # stage-level entry points (CorrespondenceFinder::compute / Linearizer::update with explicit inputs)
    def computeCorrespondences(self, referenceIndexImage, currentIndexImage, T):
        """correspondencefinder.cpp:20-118 -> (correspondences [C,2], K)"""
        p = self.params()
        ri = np.ascontiguousarray(referenceIndexImage, np.int32); ci = np.ascontiguousarray(currentIndexImage, np.int32)
        corr = np.empty((p.rows * p.cols, 2), np.int32)
        nC, nK = C.c_int(0), C.c_int(0)
        self.ctx.check(self.ctx._L.pwn_hip_correspondences(self.ctx.h, C.byref(p), self._referenceCloud.h, self._currentCloud.h, _ptr(ri), _ptr(ci),
                                                           _ptr(_colmajor(T, 4)), _ptr(corr), C.byref(nC), C.byref(nK)))
        f = self._correspondenceFinder
        f._correspondences, f._numCorrespondences = corr, nC.value
        return corr[:nC.value].copy(), nK.value

    def linearize(self, correspondences, T):
        """linearizer.cpp:17-115 with _T = T"""
        p = self.params()
        corr = np.ascontiguousarray(correspondences, np.int32)
        H = np.empty(36, np.float32); b = np.empty(6, np.float32)
        err, inl = C.c_float(0), C.c_int(0)
        self.ctx.check(self.ctx._L.pwn_hip_linearize(self.ctx.h, C.byref(p), self._referenceCloud.h, self._currentCloud.h, _ptr(corr), len(corr),
                                                     _ptr(_colmajor(T, 4)), _ptr(H), _ptr(b), C.byref(err), C.byref(inl)))
        l = self._linearizer
        l._H, l._b, l._error, l._inliers = H.reshape(6, 6).T.copy(), b, err.value, inl.value
        return dict(H=l._H, b=b, chi2=err.value, inliers=inl.value)


class PwnMatcherBase:
    """pwn_tracker/pwn_matcher_base.{h,cpp}: the caller-side boundary of the path -- makeCloud (depth image -> cloud at
    1/scale resolution) and matchClouds (align + depth-agreement score), with the reference's quirks kept:
    the initial guess' z translation is zeroed (.cpp:114), the aligner's projector is re-configured and scaled on every
    call (.cpp:117-119), the returned information matrix is the 100*I hack (.cpp:147-148)."""

    def __init__(self, aligner: "Aligner", converter: DepthImageConverterIntegralImage):
        self._aligner, self._converter = aligner, converter
        self._scale = 2                              # pwn_matcher_base.cpp:12
        self._frameInlierDepthThreshold = 50.0       # :13
        self.numCalls = 0

    def scale(self): return self._scale
    def setScale(self, s): self._scale = int(s)
    def aligner(self): return self._aligner
    def converter(self): return self._converter
    def setAligner(self, a): self._aligner = a                                                   # pwn_matcher_base.h:30,33
    def setConverter(self, c): self._converter = c

    def makeCloud(self, cameraMatrix, sensorOffset, depthImage, ctx: Context = None):
        """pwn_matcher_base.cpp:57-86 -> (cloud, r, c, scaledCameraMatrix)"""
        ctx = ctx or self._aligner.ctx
        projector = self._converter.projector()
        invScale = np.float32(1.0) / np.float32(self._scale)
        scaled = (np.asarray(cameraMatrix, np.float32).reshape(3, 3) * invScale).astype(np.float32)
        scaled[2, 2] = 1.0
        projector.setCameraMatrix(scaled)
        depth = depthImage if hasattr(depthImage, "data_ptr") else np.ascontiguousarray(depthImage, np.float32)
        rows, cols = depth.shape
        r, c = rows // self._scale, cols // self._scale
        projector.setImageSize(r, c)
        cloud = Cloud(ctx, max(1, r * c))
        # DepthImage_scale (:72) + converter->compute (:79) in one device-side call; same side effects on the projector
        projector.setTransform(np.eye(4, dtype=np.float32))
        p = self._converter.params(sensorOffset)
        ctx.check(ctx._L.pwn_hip_convert_scaled(ctx.h, C.byref(p), _ptr(depth), rows, cols, self._scale, 0.01, cloud.h))
        self.numCalls += 1
        return cloud, projector.imageRows(), projector.imageCols(), projector.cameraMatrix().copy()

    def makeCloudBegin(self, cameraMatrix, sensorOffset, depthImage, ctx: Context = None):
        """makeCloud in two halves (pwn_hip_convert_scaled_begin / pwn_hip_convert_end): returns at once with a ticket, the frame is
        converted by the library's helper thread next to whatever is run on the context meanwhile; makeCloudEnd(ticket) returns what
        makeCloud would have returned -- the same bits.  One ticket at a time per context; `depthImage` must not change in between."""
        ctx = ctx or self._aligner.ctx
        projector = self._converter.projector()
        invScale = np.float32(1.0) / np.float32(self._scale)
        scaled = (np.asarray(cameraMatrix, np.float32).reshape(3, 3) * invScale).astype(np.float32)
        scaled[2, 2] = 1.0
        projector.setCameraMatrix(scaled)
        depth = depthImage if hasattr(depthImage, "data_ptr") else np.ascontiguousarray(depthImage, np.float32)
        rows, cols = depth.shape
        r, c = rows // self._scale, cols // self._scale
        projector.setImageSize(r, c)
        cloud = Cloud(ctx, max(1, r * c))
        projector.setTransform(np.eye(4, dtype=np.float32))
        p = self._converter.params(sensorOffset)
        ctx.check(ctx._L.pwn_hip_convert_scaled_begin(ctx.h, C.byref(p), _ptr(depth), rows, cols, self._scale, 0.01, cloud.h))
        return dict(ctx=ctx, cloud=cloud, depth=depth, source=depthImage, out=(projector.imageRows(), projector.imageCols(), projector.cameraMatrix().copy()))

    def makeCloudEnd(self, ticket):
        ctx, cloud = ticket["ctx"], ticket["cloud"]
        ctx.check(ctx._L.pwn_hip_convert_end(ctx.h, cloud.h))
        self.numCalls += 1
        ticket["depth"] = None
        return (cloud,) + ticket["out"]

    def _configure(self, fromOffset, toOffset, toCameraMatrix, toRows, toCols, initialGuess):
        a = self._aligner
        projector = a.projector()
        a.setReferenceSensorOffset(fromOffset); a.setCurrentSensorOffset(toOffset)
        ig = np.asarray(np.eye(4) if initialGuess is None else initialGuess, np.float64).astype(np.float32)   # convertScalar d -> f (.h:66-71)
        ig[2, 3] = 0                                                                 # :114
        a.setInitialGuess(ig)
        projector.setCameraMatrix(toCameraMatrix); projector.setImageSize(toRows, toCols)
        projector.scale(np.float32(1.0 / self._scale))                               # :117-119 (float argument of scale())
        a.correspondenceFinder().setImageSize(projector.imageRows(), projector.imageCols())   # :128-133

    @staticmethod
    def _result(r, m: MatchResult):
        omega = np.eye(6) * 100.0                                                    # :147-148 HACK kept
        return dict(transform=r["T"].astype(np.float64), informationMatrix=omega, cloud_inliers=r["inliers"],
                    image_nonZeros=m.image_non_zeros, image_outliers=m.image_outliers, image_inliers=m.image_inliers,
                    image_reprojectionDistance=m.image_reprojection_distance, align=r)

    def matchClouds(self, fromCloud, toCloud, fromOffset, toOffset, toCameraMatrix, toRows, toCols, initialGuess=None):
        """pwn_matcher_base.cpp:88-183 -> MatcherResult as a dict"""
        a = self._aligner
        self._configure(fromOffset, toOffset, toCameraMatrix, toRows, toCols, initialGuess)
        a.setReferenceCloud(fromCloud); a.setCurrentCloud(toCloud)
        r = a.align()                                                                # :136
        m = MatchResult()
        a.ctx.check(a.ctx._L.pwn_hip_match_score(a.ctx.h, self._frameInlierDepthThreshold, C.byref(m)))    # :153-182
        return self._result(r, m)

    def matchCloudsBatch(self, fromClouds, toClouds, fromOffset, toOffset, toCameraMatrix, toRows, toCols, initialGuesses=None):
        """The candidate loop of PwnCloser::processPartition (pwn_tracker/pwn_closer.cpp:92-111) as one batched call."""
        a = self._aligner
        n = len(fromClouds)
        self._configure(fromOffset, toOffset, toCameraMatrix, toRows, toCols, None)
        p = a.params()
        res, sc = (AlignResult * n)(), (MatchResult * n)()
        refs = (C.c_void_p * n)(*[c.h for c in fromClouds]); curs = (C.c_void_p * n)(*[c.h for c in toClouds])
        gl = []
        for i in range(n):
            ig = np.asarray(np.eye(4) if initialGuesses is None else initialGuesses[i], np.float64).astype(np.float32)
            ig[2, 3] = 0; ig[3] = (0, 0, 0, 1)
            gl.append(_colmajor(ig, 4))
        g = np.ascontiguousarray(np.stack(gl), np.float32)
        a.ctx.check(a.ctx._L.pwn_hip_match_batch(a.ctx.h, C.byref(p), n, refs, curs, _ptr(g), self._frameInlierDepthThreshold, res, sc))
        return [self._result(Aligner._unpack(r), m) for r, m in zip(res, sc)]

    def matchHandles(self, fromClouds, toClouds, initialGuesses=None):
        """handle arrays + the guesses as matchClouds conditions them (z translation zeroed, pwn_matcher_base.cpp:114) for
        matchCloudsBatchRecords(..., prepared=...)"""
        n = len(fromClouds)
        gl = []
        for i in range(n):
            ig = np.asarray(np.eye(4) if initialGuesses is None else initialGuesses[i], np.float64).astype(np.float32)
            ig[2, 3] = 0; ig[3] = (0, 0, 0, 1)
            gl.append(_colmajor(ig, 4))
        g = np.ascontiguousarray(np.stack(gl), np.float32) if n else np.zeros((0, 16), np.float32)
        return ((C.c_void_p * n)(*[c.h for c in fromClouds]), (C.c_void_p * n)(*[c.h for c in toClouds]), g, n)

    def matchCloudsBatchRecords(self, fromClouds, toClouds, fromOffset, toOffset, toCameraMatrix, toRows, toCols, records, initialGuesses=None,
                                pair_ids=None, first_pair_id=0, want_results=True, prepared=None):
        """matchCloudsBatch whose results (also) leave as MATCH_RECORD_FLOATS-float records written on the device (pwn_hip_match_batch_records):
        what the ranks of a sharded PwnCloser::processPartition exchange.  records: float32 [n, 72] CUDA tensor or numpy array.  Returns
        (align results as a structured array, scores as a MatchResult array) or None."""
        a = self._aligner
        self._configure(fromOffset, toOffset, toCameraMatrix, toRows, toCols, None)
        p = a.params()
        refs, curs, g, n = prepared if prepared is not None else self.matchHandles(fromClouds, toClouds, initialGuesses)
        _check_records(records, n, MATCH_RECORD_FLOATS, a.ctx)
        ids = None if pair_ids is None else np.ascontiguousarray(pair_ids, np.int32)
        if ids is not None and ids.size < n:
            raise ValueError("pair_ids shorter than the batch")
        res = (AlignResult * n)() if want_results else None
        sc = (MatchResult * n)() if want_results else None
        a.ctx.check(a.ctx._L.pwn_hip_match_batch_records(a.ctx.h, C.byref(p), n, refs, curs, _ptr(g), self._frameInlierDepthThreshold, _ptr(ids),
                                                         int(first_pair_id), res, sc, _ptr(records)))
        return (np.frombuffer(res, dtype=ALIGN_RESULT_DTYPE, count=n), sc) if want_results else None


class PwnCloserAcceptance:
    """Acceptance rule of PwnCloser::matchFrames (pwn_tracker/pwn_closer.cpp:56-58,138-141)."""

    def __init__(self, frameMinNonZeroThreshold=3000, frameMaxOutliersThreshold=100, frameMinInliersThreshold=1000):
        self.frameMinNonZeroThreshold, self.frameMaxOutliersThreshold, self.frameMinInliersThreshold = frameMinNonZeroThreshold, frameMaxOutliersThreshold, frameMinInliersThreshold

    def accept(self, result) -> bool:
        return not (result["image_nonZeros"] < self.frameMinNonZeroThreshold or result["image_outliers"] > self.frameMaxOutliersThreshold
                    or result["image_inliers"] < self.frameMinInliersThreshold)


def iso_inverse(T):
    """Eigen::Isometry3f::inverse() in float with the CPU path's evaluation order"""
    o = np.empty(16, np.float32)
    _lib.lib().pwn_hip_iso_inverse(_ptr(_colmajor(T, 4)), _ptr(o))
    return o.reshape(4, 4).T.copy()


def iso_mul(A, B):
    """Isometry3f * Isometry3f in float with the CPU path's evaluation order"""
    o = np.empty(16, np.float32)
    _lib.lib().pwn_hip_iso_mul(_ptr(_colmajor(A, 4)), _ptr(_colmajor(B, 4)), _ptr(o))
    return o.reshape(4, 4).T.copy()


def _mul3(A, B):
    """3x3 float product with Eigen's left-to-right inner products (no BLAS, no FMA)"""
    A = np.asarray(A, np.float32); B = np.asarray(B, np.float32)
    R = np.empty((3, 3), np.float32)
    for i in range(3):
        for j in range(3):
            s = np.float32(A[i, 0] * B[0, j]); s = np.float32(s + np.float32(A[i, 1] * B[1, j])); s = np.float32(s + np.float32(A[i, 2] * B[2, j]))
            R[i, j] = s
    return R


class CloudCache:
    """Device-resident LRU of clouds keyed by frame (pwn_tracker/pwn_tracker_cache.cpp:24-51 + boss_map_building cache):
    a miss re-runs the converter on the frame's stored depth image (PwnCache::loadFrame), so loop-closure batches do not
    re-convert frames that are still resident in HBM."""

    def __init__(self, matcher: "PwnMatcherBase", capacity: int = 64):
        self._matcher, self._capacity = matcher, capacity
        self._clouds = {}        # key -> Cloud, insertion order = recency
        self._frames = {}        # key -> (depthImage, cameraMatrix, sensorOffset)
        self.hits = self.misses = 0

    def addFrame(self, key, depthImage, cameraMatrix, sensorOffset, cloud: Cloud = None):
        self._frames[key] = (np.ascontiguousarray(depthImage, np.float32), np.asarray(cameraMatrix, np.float32).copy(),
                             np.asarray(sensorOffset, np.float32).copy())
        if cloud is not None:
            self._insert(key, cloud)

    def _insert(self, key, cloud):
        self._clouds.pop(key, None)
        self._clouds[key] = cloud
        while len(self._clouds) > self._capacity:
            self._clouds.pop(next(iter(self._clouds)))          # least recently used

    def get(self, key) -> Cloud:
        c = self._clouds.pop(key, None)
        if c is not None:
            self.hits += 1
            self._clouds[key] = c
            return c
        self.misses += 1
        depth, K, off = self._frames[key]
        c, _, _, _ = self._matcher.makeCloud(K, off, depth)    # pwn_tracker_cache.cpp:38-44
        self._insert(key, c)
        return c


class PwnTracker(PwnMatcherBase):
    """pwn_tracker/pwn_tracker.{h,cpp}: sequential odometry with key-cloud switching (processFrame, .cpp:106-215).
    The BOSS map bookkeeping of the reference (frames / relations written to the map manager, :217-281) is reported
    through the returned dict instead."""

    def __init__(self, aligner, converter):
        super().__init__(aligner, converter)
        self._previousCloud = None
        self._globalT = np.eye(4, dtype=np.float32)
        self._previousCloudTransform = np.eye(4, dtype=np.float32)
        self._previousCloudOffset = np.eye(4, dtype=np.float32)
        self._newFrameInliersFraction = 0.4          # pwn_tracker.cpp:36
        self._counter = 0
        self._numKeyframes = 0
        self._ahead = None                           # prefetch(): (image, offset, K, ticket of makeCloudBegin)

    def globalT(self): return self._globalT
    def numKeyframes(self): return self._numKeyframes
    def setNewFrameInliersFraction(self, v): self._newFrameInliersFraction = float(v)
    def newFrameInliersFraction(self): return self._newFrameInliersFraction

    def init(self):
        """pwn_tracker.cpp:38-49"""
        self.dropPrefetched()
        self._previousCloud = None
        self._globalT = np.eye(4, dtype=np.float32); self._previousCloudTransform = np.eye(4, dtype=np.float32)
        self._counter = 0; self._numKeyframes = 0

    def prefetch(self, depthImage, sensorOffset, cameraMatrix):
        """Not in the reference: hand over the NEXT frame of a recorded / buffered stream before processFrame of the current one.  Its
        makeCloud (pwn_tracker.cpp:115 -- independent of the alignment of the frames before it) then runs next to that alignment;
        processFrame(depthImage, ...) of the very same image object and arguments picks the cloud up.  Same results, bit for bit."""
        self.dropPrefetched()
        off = np.asarray(sensorOffset, np.float32).copy(); Km = np.asarray(cameraMatrix, np.float32).copy()
        self._ahead = (depthImage, off, Km, self.makeCloudBegin(Km, off, depthImage))

    def dropPrefetched(self):
        if self._ahead is not None:
            ahead, self._ahead = self._ahead, None
            self.makeCloudEnd(ahead[3])

    def _currentCloud(self, cameraMatrix, sensorOffset, depthImage):
        ahead = self._ahead
        if ahead is not None and ahead[0] is depthImage and np.array_equal(ahead[1], sensorOffset) and \
                np.array_equal(ahead[2], np.asarray(cameraMatrix, np.float32)):
            self._ahead = None
            return self.makeCloudEnd(ahead[3])
        self.dropPrefetched()                                     # a different frame: the prefetched cloud is not this one's
        return self.makeCloud(cameraMatrix, sensorOffset, depthImage)

    def processFrame(self, depthImage, sensorOffset, cameraMatrix, initialGuess=None, nextDepthImage=None):
        """pwn_tracker.cpp:106-215.  nextDepthImage (not in the reference): the frame the next call will bring; it is handed to prefetch()
        once this frame's cloud exists, so that its conversion runs next to this frame's alignment."""
        a = self._aligner
        initialGuess = np.eye(4, dtype=np.float32) if initialGuess is None else np.asarray(initialGuess, np.float32)
        currentCloudOffset = np.asarray(sensorOffset, np.float32)
        currentCloud, r, c, scaledCameraMatrix = self._currentCloud(cameraMatrix, currentCloudOffset, depthImage)     # :115
        if nextDepthImage is not None:
            self.prefetch(nextDepthImage, currentCloudOffset, cameraMatrix)
        out = dict(newFrame=False, aligned=False, inliers=0, error=0.0, T=None)
        if self._previousCloud is not None:
            a.setCurrentSensorOffset(currentCloudOffset); a.setCurrentCloud(currentCloud)
            a.setReferenceSensorOffset(self._previousCloudOffset); a.setReferenceCloud(self._previousCloud)
            a.correspondenceFinder().setImageSize(r, c)
            a.projector().setCameraMatrix(scaledCameraMatrix); a.projector().setImageSize(r, c)
            guess = iso_mul(iso_mul(iso_inverse(self._previousCloudTransform), self._globalT), initialGuess)      # :132
            a.setInitialGuess(guess)
            res = a.align()                                                                                        # :136
            if res["inliers"] > 0:
                self._globalT = iso_mul(self._previousCloudTransform, res["T"])                                   # :147
            else:
                self._globalT = iso_mul(self._globalT, guess)                                                     # :150
            if not (self._counter % 50):                                                                            # :154-159
                R = self._globalT[:3, :3].astype(np.float32)
                E = _mul3(R.T, R); E[np.arange(3), np.arange(3)] -= np.float32(1)
                self._globalT[:3, :3] = (R - _mul3(np.float32(0.5) * R, E)).astype(np.float32)
            self._globalT[3] = (0, 0, 0, 1)
            inliersFraction = np.float32(res["inliers"]) / np.float32(r * c)
            out.update(aligned=True, inliers=res["inliers"], error=res["error"], T=res["T"], inliersFraction=float(inliersFraction))
            if inliersFraction < self._newFrameInliersFraction:                                                     # :164-185
                out["newFrame"] = True
                self._numKeyframes += 1
                self._previousCloud = currentCloud
                self._previousCloudTransform = self._globalT.copy()
        else:                                                                                                       # :194-200
            out["newFrame"] = True
            self._previousCloud = currentCloud
            self._previousCloudTransform = self._globalT.copy()
            self._previousCloudOffset = currentCloudOffset.copy()
            self._numKeyframes += 1
        self._counter += 1
        out["globalT"] = self._globalT.copy()
        return out


def v2t(v):
    """pwn_core/bm_se3.h:37-43"""
    v = np.ascontiguousarray(v, np.float32); T = np.empty(16, np.float32)
    _lib.lib().pwn_hip_v2t(_ptr(v), _ptr(T))
    return T.reshape(4, 4).T.copy()


def t2v(T):
    """pwn_core/bm_se3.h:45-52"""
    v = np.empty(6, np.float32)
    _lib.lib().pwn_hip_t2v(_ptr(_colmajor(T, 4)), _ptr(v))
    return v


def ldlt_solve6(H, b):
    """Matrix6f::ldlt().solve(b) (pwn_core/aligner.cpp:110): host compilation of the device function"""
    Hc = np.ascontiguousarray(np.asarray(H, np.float32).T).reshape(-1)      # column-major
    bc = np.ascontiguousarray(b, np.float32)
    x = np.empty(6, np.float32)
    _lib.lib().pwn_hip_ldlt_solve6(_ptr(Hc), _ptr(bc), _ptr(x))
    return x
