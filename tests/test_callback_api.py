"""pwn_hip_ctx_set_enqueued_callback + pwn_hip_ctx_signal_stream at the C-ABI (no torch): the callback runs once inside every alignment batch call, after the
call's device work is queued and before it waits; work the caller queues on a stream of its own behind pwn_hip_ctx_signal_stream sees the records the call
packs -- the mechanism that takes the host's share of a multi-GPU step (pwn_tracker/pwn_closer.cpp:85-111 sharded) off the critical path."""
import ctypes as C

import numpy as np
import pytest

from conftest import case_params, make_depth_pair
from test_gpu_parity import gpu_objects

pytestmark = pytest.mark.gpu


def _hip():
    h = C.CDLL("libamdhip64.so")
    h.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
    h.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    h.hipStreamSynchronize.argtypes = [C.c_void_p]
    h.hipStreamDestroy.argtypes = [C.c_void_p]
    return h


def test_callback_runs_inside_the_call_and_signal_stream_orders_a_copy_of_the_records():
    from g2o_frontend_amd import api
    hip = _hip()
    rows, cols, K, conv, alig = case_params("small")
    ref, cur, _, _, _ = make_depth_pair("small", 3)
    ctx = api.Context(0, rows, cols, 32)
    try:
        _, converter, aligner = gpu_objects(ctx, "small")
        n = 24
        refs = [api.Cloud(ctx, rows * cols) for _ in range(n)]; curs = [api.Cloud(ctx, rows * cols) for _ in range(n)]
        converter.computeBatch(refs + curs, [ref] * n + [cur] * n)
        rec = ctx.upload(np.full((n, api.RECORD_FLOATS), -3.0, np.float32))
        copy = ctx.upload(np.full((n, api.RECORD_FLOATS), -9.0, np.float32))
        s = C.c_void_p()
        assert hip.hipStreamCreateWithFlags(C.byref(s), 1) == 0                       # hipStreamNonBlocking: not ordered against anything by itself
        calls = []

        def inside():
            calls.append(len(calls))
            ctx.signal_stream(s.value)                                               # the stream continues after everything this call has queued ...
            assert hip.hipMemcpyAsync(copy.data_ptr(), rec.data_ptr(), rec.nbytes, 3, s) == 0      # ... so this device-to-device copy sees the packed records

        ctx.set_enqueued_callback(inside)
        res = aligner.alignBatchRecords(refs, curs, rec)
        ctx.take_callback_error()
        assert calls == [0]                                                          # once, during the call
        assert hip.hipStreamSynchronize(s) == 0
        a, b = rec.numpy(), copy.numpy()
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and not (a == -3.0).any() and (a[:, 63] == 0).all()
        assert np.array_equal(a[:, :16].view(np.uint32), np.ascontiguousarray(res["T"]).view(np.uint32))
        # an empty batch does not call it; an exception inside it does not cross the C frames and is handed over afterwards
        aligner.alignBatchRecords([], [], ctx.upload(np.zeros((1, api.RECORD_FLOATS), np.float32)))
        assert calls == [0]

        def boom():
            raise RuntimeError("inside the call")
        ctx.set_enqueued_callback(boom)
        aligner.alignBatch(refs[:2], curs[:2])                                       # the call itself completes
        with pytest.raises(RuntimeError, match="inside the call"):
            ctx.take_callback_error()
        ctx.set_enqueued_callback(None)
        aligner.alignBatch(refs[:2], curs[:2])
        ctx.take_callback_error()                                                    # nothing pending, nothing installed
        assert calls == [0]
        hip.hipStreamDestroy(s)
    finally:
        ctx.close()
