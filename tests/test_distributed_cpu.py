"""CPU: the N > 1 path (query shards + all-gather of the per-shard top-k) with world_size 2 on the
gloo backend.  The local search function is the oracle here; on the GPU box it is the HIP path."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from util import REPO, distinct_labels, sift_like, windows


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nq, out_dir, direct=False, weighted=False):
    import sys
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WANN_NO_TORCH="1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as orc
    from rangefilteredann_amd.distributed import shard_bounds, sharded_batch_search, weighted_bounds
    n, d = 1500, 32
    g = sift_like(n, d, 1)
    X, Q = g(n), g(nq)
    labels = distinct_labels(n, 2)
    W = windows(labels, nq, -2, 3).astype(np.float32)
    idx = orc.VamanaRangeFilterTreeIndexFloatEuclidian(X, labels, cutoff=300, split_factor=2, build_params=orc.BuildParams(16, 32, 1.0, ""), threads=2)
    qp = orc.QueryParams(10, 20)
    calls = []

    def search_plain(q, r, base):
        return search_any(q, r, base)

    def search_direct(q, r, base, out_ids, out_dists):  # (writes its rows into the all-gather's send planes)
        ids, dists = search_any(q, r, base)
        out_ids.copy_(ids)
        out_dists.copy_(dists)
        return out_ids, out_dists

    def search_any(q, r, base):
        calls.append((int(base), q.shape[0]))
        # the oracle numbers queries from 0: emulate the global numbering by searching a padded batch
        full_q = np.zeros((base + q.shape[0], d), dtype=np.float32)
        full_r = np.zeros((base + q.shape[0], 2), dtype=np.float32)
        full_q[base:], full_r[base:] = q.numpy(), r.numpy()
        full_r[:base] = (5.0, 6.0)  # empty windows: no work
        ids, dists = idx.batch_search(full_q, full_r, base + q.shape[0], "optimized_postfilter", qp)
        return torch.from_numpy(ids[base:].view(np.int32).copy()), torch.from_numpy(dists[base:].copy())

    bounds = None
    if weighted:  # a cost-balanced cut: the first queries are "heavy" -- uneven shard sizes, the same cut on every rank
        costs = np.where(np.arange(nq) < nq // 5, 40.0, 1.0) + (np.arange(nq) % 7)
        bounds = weighted_bounds(costs, world)
        assert len({b - a for a, b in bounds}) > 1
    ids, dists = sharded_batch_search(search_direct if direct else search_plain, torch.from_numpy(Q), torch.from_numpy(W), 10, bounds=bounds)
    lo, hi = bounds[rank] if bounds else shard_bounds(nq, world, rank)
    assert calls == [(lo, hi - lo)]
    eids, edists = idx.batch_search(Q, W, nq, "optimized_postfilter", qp)
    ok = np.array_equal(ids.numpy().view(np.uint32), eids) and np.array_equal(dists.numpy(), edists)
    open(os.path.join(out_dir, f"ok{rank}"), "w").write("1" if ok else "0")
    dist.destroy_process_group()


@pytest.mark.parametrize("nq,direct", [(64, False), (77, False), (64, True), (77, True)])
def test_two_rank_sharded_search_equals_single_process(oracle, tmp_path, nq, direct):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), nq, str(tmp_path), direct), nprocs=world, join=True)
    for r in range(world):
        assert open(tmp_path / f"ok{r}").read() == "1"


@pytest.mark.parametrize("world,nq,direct", [(4, 61, False), (4, 64, True), (8, 45, True)])
def test_cost_balanced_shards_on_four_and_eight_ranks(oracle, tmp_path, world, nq, direct):
    """the strong-scaling cut (weighted_bounds: contiguous shards of equal predicted work, uneven sizes, odd batch sizes) through
    the same all-gather on 4 and 8 gloo ranks: every rank ends with every row of the single-process search"""
    mp.spawn(_worker, args=(world, _free_port(), nq, str(tmp_path), direct, True), nprocs=world, join=True)
    for r in range(world):
        assert open(tmp_path / f"ok{r}").read() == "1"


def test_weighted_bounds_balance_and_cover():
    from rangefilteredann_amd.distributed import weighted_bounds
    rng = np.random.default_rng(3)
    for nq in (0, 1, 3, 8, 1000, 10001):
        for world in (1, 2, 3, 8):
            for law in ("flat", "skewed", "zeros", "one_giant"):
                c = {"flat": np.ones(nq), "skewed": rng.pareto(1.2, nq) + 0.01, "zeros": np.zeros(nq),
                     "one_giant": np.where(np.arange(nq) == nq // 2, 1e6, 1.0)}[law]
                b = weighted_bounds(c, world)
                assert len(b) == world and b[0][0] == 0 and b[-1][1] == nq
                assert all(b[i][1] == b[i + 1][0] for i in range(world - 1)) and all(hi >= lo for lo, hi in b)
                if nq >= world:
                    assert all(hi > lo for lo, hi in b), (nq, world, law, b)
                else:  # fewer queries than shards: the FIRST nq shards hold one query each, the rest are empty
                    assert b == [(r, r + 1) for r in range(nq)] + [(nq, nq)] * (world - nq), (nq, world, law, b)
                if nq and law in ("flat", "skewed"):  # no shard carries more than its share plus one query's cost
                    tot, mx = float(c.sum()), float(c.max())
                    assert max(float(c[lo:hi].sum()) for lo, hi in b) <= tot / world + mx + 1e-9
                assert b == weighted_bounds(c.astype(np.float32), world) or law == "skewed"  # (deterministic for identical inputs)
    assert weighted_bounds(c, 4) == weighted_bounds(c.copy(), 4)


def test_output_planes_are_detected_by_parameter_name():
    from rangefilteredann_amd.distributed import _takes_outputs
    assert _takes_outputs(lambda q, r, base, out_ids=None, out_dists=None: None)
    assert _takes_outputs(lambda q, r, base, out_ids, out_dists: None)
    assert not _takes_outputs(lambda q, r, base, k, stream: None)       # five parameters for other reasons
    assert not _takes_outputs(lambda q, r, base: None)
    assert not _takes_outputs(lambda q, r, base, out_ids=None: None)    # both or neither


def test_shard_bounds_cover_everything():
    from rangefilteredann_amd.distributed import shard_bounds, shard_capacity
    for nq in (0, 1, 7, 8, 10000, 10001):
        for world in (1, 2, 3, 8):
            edges = [shard_bounds(nq, world, r) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == nq
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            assert max(b - a for a, b in edges) <= shard_capacity(nq, world)


def _level_worker(rank, world, port, out_dir, kind, mult, max_beam):
    import sys
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WANN_NO_TORCH="1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as orc
    from rangefilteredann_amd.distributed import level_dealt_batch_search
    n, d, nq, k, beam = 2500, 16, 37, 10, 5
    g = sift_like(n, d, 5)
    X, Q = g(n), g(nq)
    labels = distinct_labels(n, 6)
    rng = np.random.default_rng(9)
    # windows from 2^-7 (exact scans on the tree) to 2^-1: chains of one to six levels
    W = np.concatenate([windows(labels, nq, p, 50 + p)[i::6] for i, p in enumerate((-7, -5, -4, -3, -2, -1))])[:nq].astype(np.float32)
    if kind == "tree":
        idx = orc.VamanaRangeFilterTreeIndexFloatEuclidian(X, labels, cutoff=300, split_factor=2, build_params=orc.BuildParams(16, 32, 1.0, ""), threads=2)
        search = lambda qp: idx.batch_search(Q, W, nq, "optimized_postfilter", qp)  # noqa: E731
    else:
        idx = orc.PostfilterVamanaIndexFloatEuclidian(X, filters=labels, build_params=orc.BuildParams(16, 32, 1.0, ""), threads=2)
        search = lambda qp: idx.batch_search(Q, W, nq, qp)  # noqa: E731
    cache, calls = {}, []

    def run_group(qn, b, mb, m):
        """the oracle has no per-query ids: the WHOLE batch is searched at that setting (row numbers = ids) and the rows asked for are returned"""
        calls.append((tuple(qn.tolist()), b, mb, m))
        if (b, mb, m) not in cache:
            cache[(b, mb, m)] = search(orc.QueryParams(k, b, 1.35, 10**7, 10**4, m, mb, None, False))
        ids, dists = cache[(b, mb, m)]
        sel = qn.numpy()
        return torch.from_numpy(ids[sel].view(np.int32).copy()), torch.from_numpy(dists[sel].copy())

    levels = rng.integers(1, 6, nq).tolist()  # (the same prediction on every rank; ANY values must give the same rows)
    ids, dists = level_dealt_batch_search(run_group, nq, k, beam, max_beam, mult, levels)
    eids, edists = search(orc.QueryParams(k, beam, 1.35, 10**7, 10**4, mult, max_beam, None, False))
    ok = np.array_equal(ids.numpy().view(np.uint32), eids) and np.array_equal(dists.numpy(), edists)
    # every rank searched something, and nobody searched a query's every level (the levels were dealt)
    singles = [c for c in calls if c[2] == c[1] + 1]
    ok = ok and len(calls) > 0 and (world == 1 or len(singles) > 0)
    open(os.path.join(out_dir, f"ok{rank}"), "w").write("1" if ok else "0")
    dist.destroy_process_group()


@pytest.mark.parametrize("world,kind,mult,max_beam", [(2, "tree", 1, 10000), (4, "tree", 3, 10000), (8, "post", 2, 10000), (2, "post", 1, 60), (4, "tree", 4, 100)])
def test_doubling_levels_dealt_to_ranks_give_the_single_process_rows(oracle, tmp_path, world, kind, mult, max_beam):
    """Strong scaling below the query: the doubling levels of a query's chain (src/postfilter_vamana.h:161-181: every level restarts
    from scratch) searched as items of their own by different ranks, the sequential rule applied after the all-gather, final
    re-searches (multiply > 1) and chains that outgrow their predicted levels in a second phase, loops that end at max_beam (60 /
    100: overshoot and short results) -- rows identical to the plain call on 2 / 4 / 8 gloo ranks, the oracle as search function."""
    mp.spawn(_level_worker, args=(world, _free_port(), str(tmp_path), kind, mult, max_beam), nprocs=world, join=True)
    for r in range(world):
        assert open(tmp_path / f"ok{r}").read() == "1"
