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

Shard cuts.  By default every rank gets the same NUMBER of queries (`shard_bounds`).  For one batch cut N
ways (strong scaling) the batch's time is the slowest shard's, and a query's work varies by two orders of
magnitude with its window (a window that is 1/512 of its partition needs a beam of thousands): `weighted_bounds`
cuts contiguous shards of equal predicted WORK instead (prefix sums of per-query costs, e.g.
`index.predict_costs(windows, method, query_params)` = wann_predict_costs) -- still contiguous, so global query
numbers survive.  Every rank must compute the same cut (same costs in, same bounds out: pure integer /
float64 arithmetic on identical inputs).
"""
from __future__ import annotations

import inspect
from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_bounds(nq: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced shard [lo, hi) of rank `rank`; the first nq % world shards get one more."""
    base, rem = divmod(nq, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_capacity(nq: int, world: int) -> int:
    return (nq + world - 1) // world


def weighted_bounds(costs: Sequence[float], world: int) -> List[Tuple[int, int]]:
    """`world` contiguous shards [lo, hi) of about equal total cost: shard r ends at the first query whose prefix sum reaches
    (r + 1) / world of the total.  Every shard gets at least one query while there are queries left -- so with nq < world the
    FIRST nq shards hold one query each and the last ones are empty.  Deterministic: float64 prefix sums of the given costs."""
    import numpy as np
    c = np.maximum(np.asarray(costs, dtype=np.float64), 0.0)
    nq = int(c.shape[0])
    if nq == 0:
        return [(0, 0)] * world
    if nq < world:  # one query each for the first nq shards
        return [(r, r + 1) for r in range(nq)] + [(nq, nq)] * (world - nq)
    pre = np.cumsum(c)
    total = float(pre[-1])
    if not total > 0.0:
        return [shard_bounds(nq, world, r) for r in range(world)]
    bounds, lo = [], 0
    for r in range(world):
        if r == world - 1:
            hi = nq
        else:
            hi = int(np.searchsorted(pre, total * (r + 1) / world, side="left")) + 1  # the query that crosses the target stays in r
            hi = max(hi, lo + 1)                   # at least one query ...
            hi = min(hi, nq - (world - 1 - r))     # ... and one left for each shard after this one
        hi = min(hi, nq)
        bounds.append((lo, hi))
        lo = hi
    return bounds


def sharded_batch_search(search_fn: Callable, queries: torch.Tensor, ranges: torch.Tensor, k: int,
                         pad_id: int = 0, group=None, bounds: Optional[Sequence[Tuple[int, int]]] = None,
                         writes_outputs: Optional[bool] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Search `queries` (nq, d) / `ranges` (nq, 2) -- identical on every rank -- by shards and return
    the full (nq, k) ids (int32 view of uint32) and dists on every rank.

    `search_fn(q_shard, r_shard, query_id_base[, out_ids=, out_dists=])`: a search function with parameters NAMED
    `out_ids` / `out_dists` ((m, k) int32 / float32, contiguous) is handed the all-gather's send planes by keyword and writes
    its rows straight into them (`writes_outputs` overrides the detection); one without returns `(ids, dists)` and they are
    copied there.

    `bounds`: the shard cut, one (lo, hi) per rank, contiguous and covering [0, nq) -- the SAME list on every rank (e.g.
    `weighted_bounds`); default: equal query counts."""
    grouped = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if grouped else 1
    rank = dist.get_rank(group) if grouped else 0
    nq = queries.shape[0]
    if bounds is None:
        bounds = [shard_bounds(nq, world, r) for r in range(world)]
    bounds = [(int(a), int(b)) for a, b in bounds]
    if len(bounds) != world or bounds[0][0] != 0 or bounds[-1][1] != nq or any(bounds[i][1] != bounds[i + 1][0] for i in range(world - 1)) \
            or any(b < a for a, b in bounds):
        raise ValueError(f"bounds must be {world} contiguous shards covering [0, {nq}): {bounds}")
    lo, hi = bounds[rank]
    takes_out = _takes_outputs(search_fn) if writes_outputs is None else bool(writes_outputs)
    dev = queries.device
    if not grouped:  # no process group: one process, one GPU.  (A group of ONE rank still runs the collective.)
        if takes_out:
            out_ids = torch.empty((hi - lo, k), dtype=torch.int32, device=dev)
            out_d = torch.empty((hi - lo, k), dtype=torch.float32, device=dev)
            search_fn(queries[lo:hi], ranges[lo:hi], lo, out_ids=out_ids, out_dists=out_d)
            return out_ids, out_d
        return search_fn(queries[lo:hi], ranges[lo:hi], lo)[:2]
    cap = max(b - a for a, b in bounds)
    # send = [ids plane | dists plane], each (cap, k) and contiguous: the search writes into them; shorter shards are padded
    # because all-gather needs equal sizes
    send = torch.empty((2, cap, k), dtype=torch.int32, device=dev)
    if takes_out:
        search_fn(queries[lo:hi], ranges[lo:hi], lo, out_ids=send[0, : hi - lo], out_dists=send[1, : hi - lo].view(torch.float32))
    else:
        ids, dists = search_fn(queries[lo:hi], ranges[lo:hi], lo)[:2]
        send[0, : hi - lo] = ids.view(torch.int32)
        send[1, : hi - lo] = dists.view(torch.int32)
    if hi - lo < cap:
        send[0, hi - lo:] = pad_id
        send[1, hi - lo:] = torch.tensor(torch.finfo(torch.float32).max).view(torch.int32)
    recv = torch.empty((world, 2, cap, k), dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(recv.view(world * 2 * cap, k), send.view(2 * cap, k), group=group)
    if all(b - a == cap for a, b in bounds):  # equal shards: the gathered planes ARE the result rows, rank after rank
        if world == 1:
            return recv[0, 0], recv[0, 1].view(torch.float32)
        return recv[:, 0].reshape(nq, k), recv[:, 1].reshape(nq, k).view(torch.float32)
    out_ids = torch.empty((nq, k), dtype=torch.int32, device=dev)
    out_d = torch.empty((nq, k), dtype=torch.float32, device=dev)
    for r, (a, b) in enumerate(bounds):
        out_ids[a:b] = recv[r, 0, : b - a]
        out_d[a:b] = recv[r, 1, : b - a].view(torch.float32)
    return out_ids, out_d


def _takes_outputs(fn) -> bool:
    """Does the search function declare parameters named out_ids AND out_dists?  (By NAME: a function with five parameters for
    other reasons -- k, a stream -- must not be handed tensors in those slots.)"""
    try:
        params = inspect.signature(fn).parameters
    except (TypeError, ValueError):
        return False
    return "out_ids" in params and "out_dists" in params


# ---------------------------------------------------------------------------------------------------------------------------
# Strong scaling: the doubling LEVELS of a query's chain dealt to different ranks
# ---------------------------------------------------------------------------------------------------------------------------
FLT_MAX = float(torch.finfo(torch.float32).max)


def levels_from_costs(costs: Sequence[float], beam: int, cap: int = 8) -> List[int]:
    """Doubling levels to search ahead per query from wann_predict_costs' figures (a chain that stops at level L has cost about
    beam * (2^L - 1)): a heuristic -- it decides who searches what, never a result row."""
    import math
    return [max(1, min(cap, int(math.floor(math.log2(max(float(c), float(beam)) / float(beam) + 1.0) + 1e-9)))) for c in costs]


def _deal_and_gather(items, run_group, world: int, rank: int, k: int, dev, group):
    """items: list of (query, beam, max_beam, mult), the SAME list on every rank.  Item i is searched by rank i % world after the
    items are ordered longest search first (every rank gets a fair share of every length class); run_group(query_numbers, beam, max_beam,
    mult) -> (ids, dists) serves the items of one setting.  ONE all-gather returns every item's row to every rank:
    (ids (len(items), k) int32, dists (len(items), k) float32), row i = item i.  `dev` None: the device of run_group's rows."""
    order = sorted(range(len(items)), key=lambda i: (-items[i][1], items[i][2], items[i][3], items[i][0]))
    mine = [i for pos, i in enumerate(order) if pos % world == rank]
    cap = max((len(items) + world - 1) // world, 1)
    by_setting = {}
    for slot, i in enumerate(mine):
        by_setting.setdefault(items[i][1:], []).append((slot, items[i][0]))
    send = None
    for (beam, max_beam, mult), lst in sorted(by_setting.items(), reverse=True):
        qn = torch.tensor([q for _, q in lst], dtype=torch.int64, device=dev)
        ids, dists = run_group(qn, int(beam), int(max_beam), int(mult))[:2]
        if send is None:
            dev = ids.device if dev is None else dev
            send = torch.zeros((2, cap, k), dtype=torch.int32, device=dev)
        slots = torch.tensor([s for s, _ in lst], dtype=torch.int64, device=dev)
        send[0].index_copy_(0, slots, ids.view(torch.int32).to(dev))
        send[1].index_copy_(0, slots, dists.view(torch.int32).to(dev))
    if send is None:  # (this rank was dealt nothing: more ranks than items)
        if dev is None:
            dev = torch.device("cuda", torch.cuda.current_device()) if (world > 1 and dist.get_backend(group) == "nccl") else torch.device("cpu")
        send = torch.zeros((2, cap, k), dtype=torch.int32, device=dev)
    if world == 1:
        recv = send.unsqueeze(0)
    else:
        recv = torch.empty((world, 2, cap, k), dtype=torch.int32, device=dev)
        dist.all_gather_into_tensor(recv.view(-1, k), send.view(-1, k), group=group)
    # item order[pos] sits in plane pos % world, slot pos // world
    pos_of = torch.empty(len(items), dtype=torch.int64)
    pos_of[torch.tensor(order, dtype=torch.int64)] = torch.arange(len(items), dtype=torch.int64)
    pos_of = pos_of.to(dev)
    r_of, s_of = pos_of % world, pos_of // world
    return recv[r_of, 0, s_of], recv[r_of, 1, s_of].view(torch.float32), dev


def level_dealt_batch_search(run_group: Callable, nq: int, k: int, beam: int, max_beam: int, mult: int, levels: Sequence[int], device=None,
                             group=None, method: str = "optimized_postfilter", min_query_to_bucket_ratio=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """The batch of `sharded_batch_search`, cut for STRONG scaling: what a rank takes is not a range of queries but single searches.

    The reference's doubling loop (src/postfilter_vamana.h:161-181) searches a query's partition at beam, 2 beam, 4 beam ... -- every
    level from scratch -- until a level's final beam holds k in-window entries, then optionally once more at min(beam * multiply,
    max_beam).  A query's levels are as independent as queries are, so the first `levels[q]` of them (a prediction: any values >= 1
    give the same rows) are searched AHEAD as items of their own, dealt to the ranks longest first; the sequential rule -- the first
    level with k entries decides -- is applied after ONE all-gather, and what it asks for then (the final re-search; a chain none of
    whose predicted levels found k entries carrying on) is a second, short phase.  A rank's time is no longer its longest chain.

    run_group(query_numbers (int64 tensor), beam, max_beam, multiply) -> (ids (m, k) int32, dists (m, k) float32): the engine's
    batch_search over THOSE queries of the batch -- each under its own global number (`wann_batch_search_device_ids`; the reference's
    "a query's own id is its row number" quirk, beamSearch.h:128) -- with QueryParams(k, beam, ..., multiply, max_beam).  A single
    level is run_group(q, b, b + 1, 1): with max_beam = b + 1 the loop searches once at b and neither doubles nor re-searches.

    ONLY for query classes whose batch_search is ONE post-filter chain per query: optimized_postfilter on a tree WITHOUT
    `min_query_to_bucket_ratio` (with it a query may fall back to the several-chain fenwick search, range_filter_tree.h:460-466), the
    super tree, the stand-alone post filter -- anything else raises ValueError (use `sharded_batch_search`).  Tiny windows that take
    the exact scan return the same rows at every level and settle at the first.  `device`: where the query numbers handed to run_group
    and the gathered rows live (default: the current GPU under an `nccl` group, else the device of run_group's rows -- run_group then
    receives its query numbers on the CPU).  The sequential rule runs as tensor operations: one host copy of the per-query decision."""
    if method not in ("optimized_postfilter", "super_optimized_postfilter", "postfilter", "") or min_query_to_bucket_ratio is not None:
        raise ValueError("level_dealt_batch_search serves one post-filter chain per query: not the fenwick / three_split methods and not "
                         "optimized_postfilter with min_query_to_bucket_ratio (range_filter_tree.h:460-466); use sharded_batch_search")
    grouped = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if grouped else 1
    rank = dist.get_rank(group) if grouped else 0
    if device is None and grouped and dist.get_backend(group) == "nccl":
        # (RCCL moves device memory only; run_group is handed its query numbers on this device too)
        device = torch.device("cuda", torch.cuda.current_device())
    if len(levels) != nq:
        raise ValueError("one level count per query")
    beam, max_beam, mult = int(beam), int(max_beam), int(mult)
    # levels of query q that the loop can reach at all: beam << r < max_beam
    reach = 0
    while (beam << reach) < max_beam:
        reach += 1
    nlev = [min(int(l), reach) for l in levels]
    # ---- phase A: whole chains of the queries with one predicted level, single levels of the others
    items = []
    Lmax = max([1] + nlev)
    item_of = torch.full((nq, Lmax), -1, dtype=torch.int64)  # item of (query, level); whole chains: column 0
    for q in range(nq):
        L = nlev[q]
        if L <= 1:
            item_of[q, 0] = len(items)
            items.append((q, beam, max_beam, mult))
        else:
            for r in range(L):
                item_of[q, r] = len(items)
                items.append((q, beam << r, (beam << r) + 1, 1))
    ids_a, d_a, dev = _deal_and_gather(items, run_group, world, rank, k, device, group)
    out_ids = torch.empty((nq, k), dtype=torch.int32, device=dev)
    out_d = torch.empty((nq, k), dtype=torch.float32, device=dev)
    if nq == 0:
        return out_ids, out_d
    # ---- the sequential rule (tensor operations on the gathered rows), then phase B
    item_dev = item_of.to(dev)
    nlev_t = torch.tensor(nlev, dtype=torch.int64, device=dev)
    multi = nlev_t > 1
    found_item = (d_a < FLT_MAX).sum(1) >= k                                   # per item: its final beam held k in-window entries
    found = found_item[item_dev.clamp(min=0)] & (item_dev >= 0) & multi[:, None]  # (nq, Lmax)
    any_hit = found.any(1)
    hit = torch.where(any_hit, found.to(torch.int64).argmax(1), torch.full_like(nlev_t, -1))
    # which phase-A item holds a query's row if no second phase is needed: its whole chain, its first successful level, or -- every
    # level short and the loop at its end -- its last level
    settle_level = torch.where(multi, torch.where(any_hit, hit, nlev_t - 1), torch.zeros_like(nlev_t))
    settle_item = item_dev.gather(1, settle_level[:, None]).squeeze(1)
    out_ids.copy_(ids_a.index_select(0, settle_item))
    out_d.copy_(d_a.index_select(0, settle_item))
    hit_h = hit.cpu().tolist()  # the one host copy: phase B's item list is built on the host (the same on every rank)
    items_b, target = [], []
    for q in range(nq):
        L = nlev[q]
        if L <= 1:
            continue
        if hit_h[q] >= 0:  # :173-181: the final re-search, if its beam exceeds the level's
            b = beam << hit_h[q]
            fb = min(b * mult, max_beam)
            if fb > b:
                items_b.append((q, fb, fb + 1, 1))
                target.append(q)
        else:  # every searched level was short: the loop carries on at beam << L -- if that is still below max_beam
            b = beam << L
            if b < max_beam:
                items_b.append((q, b, max_beam, mult))
                target.append(q)
            # (else the loop ends with the last level's short rows; min(b * mult, max_beam) <= b: no re-search)
    if items_b:
        ids_b, d_b, _ = _deal_and_gather(items_b, run_group, world, rank, k, dev, group)
        tq = torch.tensor(target, dtype=torch.int64, device=dev)
        out_ids.index_copy_(0, tq, ids_b)
        out_d.index_copy_(0, tq, d_b)
    return out_ids, out_d
