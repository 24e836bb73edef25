"""Worker of tests/test_distributed_gpu.py (started under torch.distributed.run with ONE rank): the product's
N > 1 path -- sharded_batch_search with the HIP search function and the all-gather on the `nccl` backend (RCCL) --
must return exactly the rows of the plain single-process call.  Also replays the 8-way shard cut on the one GPU."""
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
from rangefilteredann_amd.distributed import shard_bounds, sharded_batch_search  # noqa: E402
from util import distinct_labels, sift_like, windows  # noqa: E402


def main():
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", device_id=dev)
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    n, d, nq, k = 6000, 64, 777, 10
    g = sift_like(n, d, 2)
    X, Q = g(n), g(nq)
    labels = distinct_labels(n, 5)
    idx = wa.VamanaRangeFilterTreeIndexFloatEuclidian(X, labels, cutoff=500, split_factor=2, build_params=wa.BuildParams(32, 64, 1.0, ""))
    W = windows(labels, nq, -3, 1).astype(np.float32)
    qp = wa.QueryParams(k, 20, 1.35, 10_000_000, 10_000, 2, 10000, None, False)
    ids, dists = idx.batch_search(Q, W, nq, "optimized_postfilter", qp)
    tq, tw = torch.from_numpy(Q).to(dev), torch.from_numpy(W).to(dev)
    calls = []

    def search_fn(q, r, base):
        m = q.shape[0]
        calls.append((int(base), m))
        oi = torch.empty((m, k), dtype=torch.int32, device=dev)
        od = torch.empty((m, k), dtype=torch.float32, device=dev)
        idx.batch_search_device(q.data_ptr(), r.data_ptr(), m, base, "optimized_postfilter", qp, oi.data_ptr(), od.data_ptr(), 0)
        return oi, od

    gi, gd = sharded_batch_search(search_fn, tq, tw, k)  # RCCL all-gather runs (a group of one rank)
    torch.cuda.synchronize()
    assert calls == [(0, nq)]
    assert gi.shape == (nq, k) and gi.data_ptr() != 0
    assert np.array_equal(gi.cpu().numpy().view(np.uint32), ids), "sharded (nccl, world 1) ids differ"
    assert np.array_equal(gd.cpu().numpy(), dists), "sharded (nccl, world 1) dists differ"
    # the 8-rank cut, replayed shard by shard on this GPU: shards keep their global query numbers
    out_i = torch.empty((nq, k), dtype=torch.int32, device=dev)
    out_d = torch.empty((nq, k), dtype=torch.float32, device=dev)
    for r in range(8):
        lo, hi = shard_bounds(nq, 8, r)
        si, sd = search_fn(tq[lo:hi], tw[lo:hi], lo)
        out_i[lo:hi], out_d[lo:hi] = si, sd
    assert np.array_equal(out_i.cpu().numpy().view(np.uint32), ids) and np.array_equal(out_d.cpu().numpy(), dists)
    dist.barrier()
    dist.destroy_process_group()
    print("NCCL_WORLD1_OK")


if __name__ == "__main__":
    main()
