"""The N>1 path on CPU: world_size-2 gloo processes shard a pair list, produce records, all-gather them."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n_pairs, q):
    import torch
    import torch.distributed as dist
    from g2o_frontend_amd import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard.shard_range(n_pairs, rank, world)
    # stand-in results: a pose that encodes the global pair id (the real ones come from Aligner.alignBatch)
    res = []
    for p in mine:
        T = np.eye(4, dtype=np.float32); T[0, 3] = p; T[1, 3] = 2 * p
        res.append(dict(T=T, error=float(p) * 0.5, inliers=100 + p, iterations=10))
    local = torch.from_numpy(shard.pack_results(res, list(mine)))
    maxn = max(len(shard.shard_range(n_pairs, r, world)) for r in range(world))
    g = shard.gather_records(local, world, maxn).numpy()
    rec = shard.assemble(g, n_pairs)
    ok = all(rec[p, 12] == p and rec[p, 13] == 2 * p and rec[p, 17] == 100 + p and rec[p, 19] == p for p in range(n_pairs))
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, ok, len(mine)))


@pytest.mark.parametrize("n_pairs", [7, 8, 1])
def test_shard_and_gather_world2(n_pairs):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_pairs, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in out)
    assert sum(n for _, _, n in out) == n_pairs


def test_shard_ranges_partition():
    from g2o_frontend_amd import shard
    for n in (0, 1, 5, 8, 1024, 1000):
        for w in (1, 2, 4, 8):
            seen = []
            for r in range(w):
                rg = shard.shard_range(n, r, w)
                seen += list(rg)
                assert all(shard.owner_of(p, n, w) == r for p in rg)
            assert seen == list(range(n))
    assert len(shard.shard_range(1024, 3, 8)) == 128


def test_record_layout_carries_the_per_iteration_traces():
    """the gathered record holds {T, chi2[10], inliers[10], C[10], flags}; both packers give the same 256 bytes."""
    from g2o_frontend_amd import shard
    from g2o_frontend_amd.api import ALIGN_RESULT_DTYPE
    assert shard.RECORD_FLOATS * 4 == 256
    raw = np.zeros(2, ALIGN_RESULT_DTYPE)
    dicts = []
    for i in range(2):
        it = 10 if i == 0 else 7
        T = np.eye(4, dtype=np.float32); T[:3, 3] = (0.1 * i, -0.2, 0.3)
        raw["T"][i] = T.T.reshape(-1); raw["iterations"][i] = it; raw["n_reference"][i] = 298000 + i; raw["n_current"][i] = 297000 + i
        raw["chi2"][i, :it] = np.linspace(1e6, 3e4, it); raw["iter_inliers"][i, :it] = 200000 + np.arange(it)
        raw["iter_correspondences"][i, :it] = 201000 + np.arange(it); raw["iter_candidates"][i, :it] = 250000 + np.arange(it)
        raw["chi2"][i, it:it + 2] = 12345.0                                   # stale tail past the last iteration: must not travel
        raw["error"][i] = raw["chi2"][i, it - 1]; raw["inliers"][i] = raw["iter_inliers"][i, it - 1]
        dicts.append(dict(T=T, error=float(raw["error"][i]), inliers=int(raw["inliers"][i]), iterations=it, chi2=raw["chi2"][i, :it].copy(),
                          iter_inliers=raw["iter_inliers"][i, :it].copy(), C=raw["iter_correspondences"][i, :it].copy(), K=raw["iter_candidates"][i, :it].copy(),
                          n_reference=int(raw["n_reference"][i]), n_current=int(raw["n_current"][i])))
    a = shard.pack_results_raw(raw, [5, 6]); b = shard.pack_results(dicts, [5, 6])
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert a[1, 19] == 6 and a[1, 62] == 7 and a[1, 27] == 0 and a[1, 26] == raw["chi2"][1, 6] and a[0, 49] == 201009 and a[0, 60] == 298000
