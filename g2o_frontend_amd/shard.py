"""Multi-GPU layer of the path: independent frame pairs shard embarrassingly.

The reference runs loop-closure candidates one after another on one Aligner
(pwn_tracker/pwn_closer.cpp:92-111); they share no mutable state, so a batch of P pairs is cut into
contiguous shards, one per rank (one process per GPU), and the only exchange is a gather of the fixed-size
result records (RCCL all-gather over xGMI when the backend is "nccl"; 256 B per pair, latency-bound).
No data-path collective exists or is needed.
"""
from __future__ import annotations

import numpy as np

# One result record per pair, 64 floats = 256 bytes ({T, chi2[10], inliers[10], C[10], flags}):
#   [0:16]  T, column-major 4x4                      [16] chi2 of the last iteration   [17] inliers   [18] iterations   [19] global pair id
#   [20:30] chi2 per iteration   [30:40] inliers per iteration   [40:50] correspondences C_i   [50:60] candidates K_i   (first TRACE iterations)
#   [60] points of the reference cloud   [61] points of the current cloud   [62] iterations carried in the traces
#   [63] 0 (1 = records of a call that is being repeated, see include/pwn_hip.h)
# Integers travel as float32 (exact: every count is < 2^24).
TRACE = 10
RECORD_FLOATS = 64


def shard_range(n_pairs: int, rank: int, world: int) -> range:
    """Contiguous shard of `rank`: pair p belongs to rank floor(p * world / n_pairs)."""
    lo = (rank * n_pairs + world - 1) // world
    hi = ((rank + 1) * n_pairs + world - 1) // world
    return range(lo, hi)


def owner_of(pair: int, n_pairs: int, world: int) -> int:
    return (pair * world) // n_pairs


def pack_results(results, pair_ids) -> np.ndarray:
    """[n, RECORD_FLOATS] float32 records from Aligner.alignBatch results (list of dicts)."""
    out = np.zeros((len(results), RECORD_FLOATS), np.float32)
    for i, (r, pid) in enumerate(zip(results, pair_ids)):
        out[i, :16] = np.asarray(r["T"], np.float32).T.reshape(-1)
        out[i, 16] = r["error"]; out[i, 17] = r["inliers"]; out[i, 18] = r["iterations"]; out[i, 19] = pid
        m = min(int(r["iterations"]), TRACE)
        for base, key in ((20, "chi2"), (30, "iter_inliers"), (40, "C"), (50, "K")):
            if key in r:
                out[i, base:base + m] = np.asarray(r[key], np.float32)[:m]
        out[i, 60] = r.get("n_reference", 0); out[i, 61] = r.get("n_current", 0); out[i, 62] = m
    return out


def pack_results_raw(results: np.ndarray, pair_ids) -> np.ndarray:
    """Same records from the structured-array form (api.ALIGN_RESULT_DTYPE; T is already column-major)."""
    n = len(results)
    out = np.zeros((n, RECORD_FLOATS), np.float32)
    out[:, :16] = results["T"]
    out[:, 16] = results["error"]; out[:, 17] = results["inliers"]; out[:, 18] = results["iterations"]; out[:, 19] = np.asarray(pair_ids, np.float32)
    m = np.minimum(results["iterations"], TRACE).astype(np.int64)
    mask = np.arange(TRACE)[None, :] < m[:, None]                       # entries past a pair's last iteration stay 0
    out[:, 20:30] = np.where(mask, results["chi2"][:, :TRACE], 0)
    out[:, 30:40] = np.where(mask, results["iter_inliers"][:, :TRACE], 0)
    out[:, 40:50] = np.where(mask, results["iter_correspondences"][:, :TRACE], 0)
    out[:, 50:60] = np.where(mask, results["iter_candidates"][:, :TRACE], 0)
    out[:, 60] = results["n_reference"]; out[:, 61] = results["n_current"]; out[:, 62] = m
    return out


def all_gather_into(out: "torch.Tensor", local: "torch.Tensor"):
    """dist.all_gather_into_tensor; with the gloo backend (CPU tests, the one-device rehearsal) the list form on views of `out`, which gloo
    implements for host and device tensors alike"""
    import torch.distributed as dist
    if dist.get_backend() == "gloo":
        dist.all_gather(list(out.view(dist.get_world_size(), *local.shape).unbind(0)), local.contiguous())
    else:
        dist.all_gather_into_tensor(out, local)


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
    all_gather_into(out, pad)
    return out


def assemble(gathered: np.ndarray, n_pairs: int) -> np.ndarray:
    """Order the gathered records by global pair id; every pair must appear exactly once."""
    ids = gathered[:, 19].astype(np.int64)
    keep = ids >= 0
    rec, ids = gathered[keep], ids[keep]
    if len(ids) != n_pairs or len(np.unique(ids)) != n_pairs:
        raise RuntimeError(f"gather incomplete: {len(ids)} records for {n_pairs} pairs")
    if rec.shape[1] >= RECORD_FLOATS and np.any(rec[:, 63] != 0):
        raise RuntimeError("records of a call that was being repeated (word 63): taken off the device before the call returned")
    return rec[np.argsort(ids, kind="stable")]
