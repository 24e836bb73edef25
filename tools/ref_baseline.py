"""CPU baseline worker for bench.py: times the REAL reference (oracle/_ref build; falls back to the oracle
port) on the bench's batch with a given PARLAY_NUM_THREADS, in its own process (the reference reads the
thread count once).  Loads the graphs bench.py left in the cache; prints one JSON line."""
import argparse, json, os, sys, time

ap = argparse.ArgumentParser()
ap.add_argument("--threads", type=int, required=True)
ap.add_argument("--n", type=int, required=True)
ap.add_argument("--nq", type=int, required=True)
ap.add_argument("--dim", type=int, required=True)
ap.add_argument("--beam", type=int, required=True)
ap.add_argument("--mult", type=int, required=True)
ap.add_argument("--cache", required=True)
ap.add_argument("--result", required=True, help="npz with the GPU ids / dists / windows of the same batch")
ap.add_argument("--seconds", type=float, default=8.0)
args = ap.parse_args()
os.environ["PARLAY_NUM_THREADS"] = str(args.threads)
os.environ["WANN_NO_TORCH"] = "1"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import bench
from oracle import oracle as orc
from util import quiet_stdout

X, Q, labels = bench.make_data(args.n, args.dim, args.nq, 0)
res = np.load(args.result)
W = res["W"].astype(np.float64)
ref = orc.load_reference(prefer=("x86-64-v4", "native", "x86-64-v3"))
R, L, alpha, cutoff, split = 64, 500, 1.0, 1000, 2
if ref is not None:
    kind, mod = "reference", ref
    with quiet_stdout():
        idx = ref.VamanaRangeFilterTreeIndexFloatEuclidian(X, labels, cutoff=cutoff, split_factor=split, build_params=ref.BuildParams(R, L, alpha, args.cache))
else:
    kind, mod = "port", orc
    idx = orc.VamanaRangeFilterTreeIndexFloatEuclidian(X, labels, cutoff=cutoff, split_factor=split, build_params=orc.BuildParams(R, L, alpha, args.cache), threads=args.threads)
qp = mod.QueryParams(10, args.beam, 1.35, 10_000_000, 10_000, args.mult, 10000, None, False)
best, reps, t_all = None, 0, time.perf_counter()
while reps < 3 or (time.perf_counter() - t_all < args.seconds and reps < 50):
    t = time.perf_counter()
    with quiet_stdout():
        ids, dists = idx.batch_search(Q, W, args.nq, "optimized_postfilter", qp)
    dt = time.perf_counter() - t
    best = dt if best is None else min(best, dt)
    reps += 1
print(json.dumps(dict(kind=kind, threads=args.threads, qps=args.nq / best, reps=reps,
                      same_ids=float((ids == res["ids"]).all(axis=1).mean()), same_dists=float((dists == res["dists"]).all(axis=1).mean()))))
