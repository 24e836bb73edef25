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
    the full (nq, k) ids (int32 view of uint32) and dists on every rank.

    `search_fn(q_shard, r_shard, query_id_base[, out_ids, out_dists])`: a search function that takes the two output
    tensors ((m, k) int32 / float32, contiguous) writes its rows straight into the all-gather's send buffer; one that
    does not returns `(ids, dists)` and they are copied there."""
    grouped = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if grouped else 1
    rank = dist.get_rank(group) if grouped else 0
    nq = queries.shape[0]
    lo, hi = shard_bounds(nq, world, rank)
    takes_out = _takes_outputs(search_fn)
    if not grouped:  # no process group: one process, one GPU.  (A group of ONE rank still runs the collective.)
        return search_fn(queries[lo:hi], ranges[lo:hi], lo)[:2]
    cap = shard_capacity(nq, world)
    dev = queries.device
    # send = [ids plane | dists plane], each (cap, k) and contiguous: the search writes into them; short shards (at most one
    # row short) are padded because all-gather needs equal sizes
    send = torch.empty((2, cap, k), dtype=torch.int32, device=dev)
    if takes_out:
        search_fn(queries[lo:hi], ranges[lo:hi], lo, send[0, : hi - lo], send[1, : hi - lo].view(torch.float32))
    else:
        ids, dists = search_fn(queries[lo:hi], ranges[lo:hi], lo)
        send[0, : hi - lo] = ids.view(torch.int32)
        send[1, : hi - lo] = dists.view(torch.int32)
    if hi - lo < cap:
        send[0, hi - lo:] = pad_id
        send[1, hi - lo:] = torch.tensor(torch.finfo(torch.float32).max).view(torch.int32)
    recv = torch.empty((world, 2, cap, k), dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(recv.view(world * 2 * cap, k), send.view(2 * cap, k), group=group)
    if nq == world * cap:  # equal shards: the gathered planes ARE the result rows, rank after rank
        if world == 1:
            return recv[0, 0], recv[0, 1].view(torch.float32)
        return recv[:, 0].reshape(nq, k), recv[:, 1].reshape(nq, k).view(torch.float32)
    out_ids = torch.empty((nq, k), dtype=torch.int32, device=dev)
    out_d = torch.empty((nq, k), dtype=torch.float32, device=dev)
    for r in range(world):
        a, b = shard_bounds(nq, world, r)
        out_ids[a:b] = recv[r, 0, : b - a]
        out_d[a:b] = recv[r, 1, : b - a].view(torch.float32)
    return out_ids, out_d


def _takes_outputs(fn) -> bool:
    import inspect
    try:
        return len(inspect.signature(fn).parameters) >= 5
    except (TypeError, ValueError):
        return False
