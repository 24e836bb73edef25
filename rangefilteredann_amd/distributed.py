"""One-process-per-GPU query sharding for batch_search.

The reference is single-process shared memory (parlay::parallel_for over queries,
src/range_filter_tree.h:70); queries are independent, so the multi-GPU form is: the index is
replicated in every GPU's HBM, the query batch is cut into contiguous shards (a shard keeps its
GLOBAL row numbers because the reference uses the query's row number as its "own id",
beamSearch.h:128 + range_filter_tree.h:71-72), every rank searches its shard with the HIP kernels
and the per-shard top-k (ids uint32, dists float32) are exchanged with ONE all-gather over
RCCL/xGMI (torch.distributed backend "nccl" on ROCm) -- 8 bytes * k per query, latency bound.

`search_fn(q_shard, r_shard, query_id_base) -> (ids, dists)` does the local search, so the same
code runs on CPU tensors under the gloo backend in the tests.
"""
from __future__ import annotations

from typing import Callable, Tuple

import torch
import torch.distributed as dist


def shard_bounds(nq: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced shard [lo, hi) of rank `rank`; the first nq % world shards get one more."""
    base, rem = divmod(nq, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_capacity(nq: int, world: int) -> int:
    return (nq + world - 1) // world


def sharded_batch_search(search_fn: Callable, queries: torch.Tensor, ranges: torch.Tensor, k: int,
                         pad_id: int = 0, group=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Search `queries` (nq, d) / `ranges` (nq, 2) -- identical on every rank -- by shards and return
    the full (nq, k) ids (int32 view of uint32) and dists on every rank."""
    grouped = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if grouped else 1
    rank = dist.get_rank(group) if grouped else 0
    nq = queries.shape[0]
    lo, hi = shard_bounds(nq, world, rank)
    ids, dists = search_fn(queries[lo:hi], ranges[lo:hi], lo)
    if not grouped:  # no process group: one process, one GPU.  (A group of ONE rank still runs the collective.)
        return ids, dists
    cap = shard_capacity(nq, world)
    # all-gather needs equal sizes: pad the short shards (at most one row)
    send = torch.empty((cap, k, 2), dtype=torch.int32, device=ids.device)
    send[: hi - lo, :, 0] = ids.view(torch.int32)
    send[: hi - lo, :, 1] = dists.view(torch.int32)
    if hi - lo < cap:
        send[hi - lo:, :, 0] = pad_id
        send[hi - lo:, :, 1] = torch.tensor(torch.finfo(torch.float32).max).view(torch.int32)
    recv = torch.empty((world, cap, k, 2), dtype=torch.int32, device=ids.device)
    dist.all_gather_into_tensor(recv.view(world * cap, k, 2), send, group=group)
    out_ids = torch.empty((nq, k), dtype=torch.int32, device=ids.device)
    out_d = torch.empty((nq, k), dtype=torch.float32, device=ids.device)
    for r in range(world):
        a, b = shard_bounds(nq, world, r)
        out_ids[a:b] = recv[r, : b - a, :, 0]
        out_d[a:b] = recv[r, : b - a, :, 1].view(torch.float32)
    return out_ids, out_d
