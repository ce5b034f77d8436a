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
