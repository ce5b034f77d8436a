"""Multi-GPU layer of the path: independent frame pairs shard embarrassingly.

The reference runs loop-closure candidates one after another on one Aligner
(pwn_tracker/pwn_closer.cpp:92-111); they share no mutable state, so a batch of P pairs is cut into
contiguous shards, one per rank (one process per GPU), and the only exchange is a gather of the fixed-size
result records (RCCL all-gather over xGMI when the backend is "nccl"; ~100 B per pair, latency-bound).
No data-path collective exists or is needed.
"""
from __future__ import annotations

import numpy as np

RECORD_FLOATS = 16 + 4          # pose (column-major 4x4) + chi2, inliers, iterations, global pair id


def shard_range(n_pairs: int, rank: int, world: int) -> range:
    """Contiguous shard of `rank`: pair p belongs to rank floor(p * world / n_pairs)."""
    lo = (rank * n_pairs + world - 1) // world
    hi = ((rank + 1) * n_pairs + world - 1) // world
    return range(lo, hi)


def owner_of(pair: int, n_pairs: int, world: int) -> int:
    return (pair * world) // n_pairs


def pack_results(results, pair_ids) -> np.ndarray:
    """[n, RECORD_FLOATS] float32 records from Aligner.alignBatch results."""
    out = np.zeros((len(results), RECORD_FLOATS), np.float32)
    for i, (r, pid) in enumerate(zip(results, pair_ids)):
        out[i, :16] = np.asarray(r["T"], np.float32).T.reshape(-1)
        out[i, 16] = r["error"]; out[i, 17] = r["inliers"]; out[i, 18] = r["iterations"]; out[i, 19] = pid
    return out


def pack_results_raw(results: np.ndarray, pair_ids) -> np.ndarray:
    """Same records from the structured-array form (api.ALIGN_RESULT_DTYPE; T is already column-major)."""
    out = np.empty((len(results), RECORD_FLOATS), np.float32)
    out[:, :16] = results["T"]
    out[:, 16] = results["error"]; out[:, 17] = results["inliers"]; out[:, 18] = results["iterations"]; out[:, 19] = np.asarray(pair_ids, np.float32)
    return out


def gather_records(local: "torch.Tensor", world: int, max_per_rank: int, force: bool = False):
    """All-gather of the per-rank record blocks (padded to max_per_rank rows); returns [world*max_per_rank, R].
    Rows whose pair id (column 19) is negative are padding."""
    import torch
    import torch.distributed as dist
    if local.shape[0] == max_per_rank:
        pad = local
    else:
        pad = torch.full((max_per_rank, local.shape[1]), -1.0, dtype=local.dtype, device=local.device)
        pad[: local.shape[0]] = local
    if world == 1 and not force:
        return pad
    out = torch.empty((world * max_per_rank, local.shape[1]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad)
    return out


def assemble(gathered: np.ndarray, n_pairs: int) -> np.ndarray:
    """Order the gathered records by global pair id; every pair must appear exactly once."""
    ids = gathered[:, 19].astype(np.int64)
    keep = ids >= 0
    rec, ids = gathered[keep], ids[keep]
    if len(ids) != n_pairs or len(np.unique(ids)) != n_pairs:
        raise RuntimeError(f"gather incomplete: {len(ids)} records for {n_pairs} pairs")
    return rec[np.argsort(ids, kind="stable")]
