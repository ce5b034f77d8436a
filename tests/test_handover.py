"""The strip hand-over of the converter's integral-image kernels (k_unproject_integral, k_unproject_integral_rows: a strip continues the
running sums of the strip on its left through epoch-tagged words, polled with a bound).  Fault injection: a withheld word must end in
PWN_HIP_ERR_LAUNCH -- not in a hang, not in silently wrong planes -- and leave the context usable.  Soak: two contexts converting from two
host threads while a third aligns, i.e. foreign kernels occupying the CUs; the fault flag must never fire and every result must be the
bit-exact one."""
import threading

import numpy as np
import pytest

from conftest import case_params

pytestmark = pytest.mark.gpu


def _objects(ctx, name):
    from test_gpu_parity import gpu_objects
    return gpu_objects(ctx, name)


def _digest(cloud):
    a = cloud.arrays()
    return tuple(int(np.frombuffer(a[k].tobytes(), np.uint8).astype(np.uint64).sum()) ^ len(a[k]) for k in ("points", "normals", "curvature", "omega_p"))


@pytest.mark.parametrize("n_frames", [2, 20])            # 2: tiled row kernel + column kernel; 20: single-pass strip kernel
def test_withheld_handover_word_is_an_error_not_a_hang(oracle, n_frames):
    import ctypes as C
    from g2o_frontend_amd import api, synth
    from g2o_frontend_amd._lib import PwnHipError
    rows, cols, K, conv, _ = case_params("small")
    ctx = api.Context(0, rows, cols, 32)
    _, converter, _ = _objects(ctx, "small")
    frames = [synth.render_depth_mm(50 + k, np.eye(4), rows, cols, K) for k in range(n_frames)]
    clouds = [api.Cloud(ctx, rows * cols) for _ in range(n_frames)]
    converter.computeBatch(clouds, frames, raw_scale=0.001)
    good = [_digest(c) for c in clouds]
    o, _, _ = oracle.convert(oracle.converter_params(K=K, **conv), oracle.convert_16u_to_32f(frames[0]))
    assert np.array_equal(clouds[0].arrays()["normals"].view(np.uint32), o.arrays()["normals"].view(np.uint32))
    # withhold the word strip 0 -> strip 1, band 2, chain 37; 4096 polls instead of 2^20 keep the test short
    ctx.check(ctx._L.pwn_hip_debug_withhold_carry(ctx.h, 0, 2, 37, rows, 4096))
    with pytest.raises(PwnHipError) as e:
        converter.computeBatch(clouds, frames, raw_scale=0.001)
    assert e.value.code == 5 and "hand-over" in str(e.value)
    # a second faulting call behaves the same (the flag was reset), then the hook goes off and the context works as before
    with pytest.raises(PwnHipError):
        converter.computeBatch(clouds, frames, raw_scale=0.001)
    ctx.check(ctx._L.pwn_hip_debug_withhold_carry(ctx.h, -1, 0, 0, rows, 0))
    converter.computeBatch(clouds, frames, raw_scale=0.001)
    assert [_digest(c) for c in clouds] == good
    ctx.close()


@pytest.mark.parametrize("n_frames", [2, 20])
def test_a_single_timed_out_launch_is_repeated_and_the_call_succeeds(n_frames):
    """a launch whose hand-over timed out (a neighbour strip's workgroup not dispatched in time on a device shared with foreign kernels) leaves
    nothing behind -- the words are epoch-tagged --, so the library repeats the call once on its own: with a ONE-SHOT withheld word the convert
    call and the one-submission step return normally with the right results, and the repeat is counted"""
    import ctypes as C
    from g2o_frontend_amd import api, synth
    rows, cols, K, conv, _ = case_params("small")
    ctx = api.Context(0, rows, cols, 64)
    _, converter, aligner = _objects(ctx, "small")
    frames = [synth.render_depth_mm(50 + k, np.eye(4), rows, cols, K) for k in range(n_frames)]
    clouds = [api.Cloud(ctx, rows * cols) for _ in range(n_frames)]
    converter.computeBatch(clouds, frames, raw_scale=0.001)
    good = [_digest(c) for c in clouds]
    n = C.c_int(-1)
    ctx.check(ctx._L.pwn_hip_debug_convert_retries(ctx.h, C.byref(n))); assert n.value == 0
    ctx.check(ctx._L.pwn_hip_debug_withhold_carry(ctx.h, 0, 2, 37, rows, -4096))          # one disturbed launch
    converter.computeBatch(clouds, frames, raw_scale=0.001)                                # no exception
    assert [_digest(c) for c in clouds] == good
    ctx.check(ctx._L.pwn_hip_debug_convert_retries(ctx.h, C.byref(n))); assert n.value == 1
    # the one-submission step: same recovery
    h = n_frames // 2
    want = aligner.alignBatch(clouds[:h], clouds[h:], raw=True).copy()
    ctx.check(ctx._L.pwn_hip_debug_withhold_carry(ctx.h, 0, 1, 5, rows, -4096))
    got = aligner.convertAlignBatch(converter, clouds[:h], clouds[h:], frames[:h], frames[h:], raw_scale=0.001)
    assert np.array_equal(got["chi2"].view(np.uint32), want["chi2"].view(np.uint32)) and np.array_equal(got["T"].view(np.uint32), want["T"].view(np.uint32))
    ctx.check(ctx._L.pwn_hip_debug_convert_retries(ctx.h, C.byref(n))); assert n.value == 2
    converter.computeBatch(clouds, frames, raw_scale=0.001)                                # undisturbed again
    ctx.check(ctx._L.pwn_hip_debug_convert_retries(ctx.h, C.byref(n))); assert n.value == 2
    ctx.close()


def test_two_converting_contexts_and_an_aligner_soak(oracle):
    """40 rounds x (2 x 24-frame single-pass conversions on two contexts / host threads) next to a thread that keeps aligning on a third
    context: every conversion bit-identical to the first one, no hand-over time-out, the alignments unchanged."""
    from g2o_frontend_amd import api, synth
    rows, cols, K, conv, alig = case_params("vga")
    n, rounds = 24, 40
    frames = [synth.render_depth_mm(70 + k, np.eye(4), rows, cols, K) for k in range(4)]
    errors, digests = [], {0: [], 1: []}
    stop = threading.Event()

    def converter_thread(tid):
        try:
            ctx = api.Context(0, rows, cols, 32)
            _, converter, _ = _objects(ctx, "vga")
            clouds = [api.Cloud(ctx, rows * cols) for _ in range(n)]
            mine = [frames[(k + tid) % 4] for k in range(n)]
            for _ in range(rounds):
                converter.computeBatch(clouds, mine, raw_scale=0.001)                  # raises on a hand-over time-out
                digests[tid].append((_digest(clouds[0]), _digest(clouds[n - 1]), sum(len(c) for c in clouds)))
            ctx.close()
        except Exception as e:      # noqa: BLE001
            errors.append((tid, repr(e)))

    def aligner_thread():
        try:
            ctx = api.Context(0, rows, cols, 16)
            _, converter, aligner = _objects(ctx, "vga")
            ref_mm, cur_mm, _ = synth.make_pair(0, rows, cols, K)
            refs = [api.Cloud(ctx, rows * cols) for _ in range(8)]; curs = [api.Cloud(ctx, rows * cols) for _ in range(8)]
            converter.computeBatch(refs + curs, [ref_mm] * 8 + [cur_mm] * 8, raw_scale=0.001)
            first = None
            while not stop.is_set():
                r = aligner.alignBatch(refs, curs, raw=True)
                key = (r["T"].tobytes(), r["chi2"].tobytes())
                if first is None:
                    first = key
                elif key != first:
                    errors.append(("aligner", "result changed under load"))
                    break
            ctx.close()
        except Exception as e:      # noqa: BLE001
            errors.append(("aligner", repr(e)))

    ta = threading.Thread(target=aligner_thread)
    tc = [threading.Thread(target=converter_thread, args=(i,)) for i in range(2)]
    ta.start()
    for t in tc:
        t.start()
    for t in tc:
        t.join(timeout=600)
    stop.set()
    ta.join(timeout=120)
    assert not errors, errors
    for tid in (0, 1):
        assert len(digests[tid]) == rounds and all(d == digests[tid][0] for d in digests[tid])
    # the two contexts converted the same frames in a different order: frame (k + tid) % 4, so thread 1's first cloud is thread 0's second ...
    o, _, _ = oracle.convert(oracle.converter_params(K=K, **conv), oracle.convert_16u_to_32f(frames[0]))
    ctx = api.Context(0, rows, cols, 2)
    _, converter, _ = _objects(ctx, "vga")
    c = api.Cloud(ctx, rows * cols)
    converter.computeBatch([c, api.Cloud(ctx, rows * cols)], [frames[0], frames[1]], raw_scale=0.001)
    assert _digest(c) == digests[0][0][0]
    assert np.array_equal(c.arrays()["normals"].view(np.uint32), o.arrays()["normals"].view(np.uint32))
    ctx.close()
