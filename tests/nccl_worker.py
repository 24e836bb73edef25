"""Worker of tests/test_distributed_gpu.py, started under torch.distributed.run with N ranks (one per GPU; N = 1 on a one-GPU box):
the product's N > 1 path on the `nccl` backend (RCCL) with the HIP engine on every rank --
  * sharded_batch_search (equal-count and cost-balanced contiguous shards, ONE all-gather) and
  * level_dealt_batch_search (single doubling levels dealt to the ranks, wann_batch_search_device_ids)
must return exactly the rows of the plain single-GPU call, which every rank computes for itself on its own replica.
Usage: nccl_worker.py <expected world size>"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import rangefilteredann_amd  # noqa: E402,F401
import window_ann as wa  # noqa: E402
from rangefilteredann_amd.distributed import (level_dealt_batch_search, levels_from_costs, shard_bounds, sharded_batch_search,  # noqa: E402
                                              weighted_bounds)
from util import distinct_labels, sift_like, windows  # noqa: E402


def main():
    want_world = int(sys.argv[1])
    local = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ["WANN_DEVICE"] = str(local)
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", device_id=dev)
    world, rank = dist.get_world_size(), dist.get_rank()
    assert dist.get_backend() == "nccl" and world == want_world, (dist.get_backend(), world, want_world)  # RCCL really sees N ranks
    n, d, nq, k = 6000, 64, 777, 10
    g = sift_like(n, d, 2)
    X, Q = g(n), g(nq)
    labels = distinct_labels(n, 5)
    # every rank builds its own replica on its own GPU (the build is deterministic: identical graphs)
    idx = wa.VamanaRangeFilterTreeIndexFloatEuclidian(X, labels, cutoff=500, split_factor=2, build_params=wa.BuildParams(32, 64, 1.0, ""))
    method = "optimized_postfilter"
    ok = True
    for p, beam, mult in ((-3, 20, 2), (-7, 10, 2)):  # (2^-7 of 6000 points: windows far below their partitions -- chains of several levels)
        W = windows(labels, nq, p, 1).astype(np.float32)
        qp = wa.QueryParams(k, beam, 1.35, 10_000_000, 10_000, mult, 10000, None, False)
        ids, dists = idx.batch_search(Q, W, nq, method, qp)  # the single-GPU call, on this rank's replica
        tq, tw = torch.from_numpy(Q).to(dev), torch.from_numpy(W).to(dev)
        calls = []

        def search_fn(q, r, base):
            m = q.shape[0]
            calls.append((int(base), m))
            oi = torch.empty((m, k), dtype=torch.int32, device=dev)
            od = torch.empty((m, k), dtype=torch.float32, device=dev)
            idx.batch_search_device(q.data_ptr(), r.data_ptr(), m, base, method, qp, oi.data_ptr(), od.data_ptr(), 0)
            return oi, od

        def same(gi, gd, what):
            torch.cuda.synchronize()
            a = np.array_equal(gi.cpu().numpy().view(np.uint32), ids) and np.array_equal(gd.cpu().numpy(), dists)
            if not a:
                print(f"[rank {rank}] {what} at 2^{p}: rows differ from the single-GPU call", flush=True)
            return a

        gi, gd = sharded_batch_search(search_fn, tq, tw, k)  # RCCL all-gather (also in a group of one rank)
        lo, hi = shard_bounds(nq, world, rank)
        assert calls == [(lo, hi - lo)], calls  # this rank searched exactly its shard, under its global query numbers
        assert gi.shape == (nq, k)
        ok &= same(gi, gd, "sharded_batch_search (equal counts)")
        costs = idx.predict_costs(W, method, qp)
        gi, gd = sharded_batch_search(search_fn, tq, tw, k, bounds=weighted_bounds(costs, world))
        ok &= same(gi, gd, "sharded_batch_search (cost-balanced shards)")

        def run_group(qn, b, mb, m):
            qs, ws = tq[qn].contiguous(), tw[qn].contiguous()
            ri = torch.empty((qn.shape[0], k), dtype=torch.int32, device=dev)
            rd = torch.empty((qn.shape[0], k), dtype=torch.float32, device=dev)
            idx.batch_search_device_ids(qs.data_ptr(), ws.data_ptr(), qn.shape[0], qn.contiguous().data_ptr(), method,
                                        wa.QueryParams(k, b, 1.35, 10_000_000, 10_000, m, mb, None, False), ri.data_ptr(), rd.data_ptr(), 0)
            return ri, rd
        gi, gd = level_dealt_batch_search(run_group, nq, k, beam, 10000, mult, levels_from_costs(costs, beam), device=dev, method=method)
        ok &= same(gi, gd, "level_dealt_batch_search")
    if world == 1:
        # the 8-rank cut, replayed shard by shard on this GPU: shards keep their global query numbers
        out_i = torch.empty((nq, k), dtype=torch.int32, device=dev)
        out_d = torch.empty((nq, k), dtype=torch.float32, device=dev)
        for r in range(8):
            lo, hi = shard_bounds(nq, 8, r)
            si, sd = search_fn(tq[lo:hi], tw[lo:hi], lo)
            out_i[lo:hi], out_d[lo:hi] = si, sd
        ok &= same(out_i, out_d, "8-way cut replayed on one GPU")
    flag = torch.tensor([int(ok)], device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dist.barrier()
    dist.destroy_process_group()
    assert int(flag.item()) == 1, "some rank saw differing rows"
    if rank == 0:
        print(f"NCCL_WORKER_OK world={world}")


if __name__ == "__main__":
    main()
