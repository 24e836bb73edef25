"""CPU-baseline worker of bench.py: the REAL reference (oracle/_ref build; falls back to the oracle port) on the bench's index --
loaded ONCE from the graph files bench.py left in the cache -- answering any number of "legs" (a window set + a (beam, mult)
setting + the GPU's rows for them) with one PARLAY_NUM_THREADS (the reference fixes its thread count at first use, so every
thread count is its own process).  Prints one JSON object: {leg name: {qps, reps, same_dists, same_ids, same_id_sets}}.

  python tools/ref_legs.py --workload sift|deep --threads T --n N --nq NQ --dim D --cache DIR --legs legs.npz [--seconds S]

legs.npz holds, per leg name L:  W|L (nq, 2) windows, set|L (beam, mult), ids|L / dists|L the GPU rows."""
import argparse, json, os, sys, time

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="sift")
ap.add_argument("--threads", type=int, required=True)
ap.add_argument("--n", type=int, required=True)
ap.add_argument("--nq", type=int, required=True)
ap.add_argument("--dim", type=int, required=True)
ap.add_argument("--cache", required=True)
ap.add_argument("--legs", required=True)
ap.add_argument("--seconds", type=float, default=3.0, help="time budget per leg (at least two calls each)")
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

wl = bench.WORKLOADS[args.workload]
X, Q, labels = wl["make"](args.n, args.dim, args.nq, 0)
legs = np.load(args.legs)
names = sorted({k.split("|", 1)[1] for k in legs.files if k.startswith("W|")})
ref = orc.load_reference(prefer=("x86-64-v4", "native", "x86-64-v3"))
kind, mod = ("reference", ref) if ref is not None else ("port", orc)
t0 = time.time()
kw = dict(cutoff=wl["cutoff"], split_factor=wl["split"], build_params=mod.BuildParams(wl["R"], wl["L"], wl["alpha"], args.cache))
if ref is None:
    kw["threads"] = args.threads
with quiet_stdout():
    idx = getattr(mod, wl["cls"])(X, labels, **kw)
load_s = time.time() - t0
out = {"_kind": kind, "_threads": args.threads, "_index_load_s": round(load_s, 1)}
for name in names:
    beam, mult = (int(x) for x in legs["set|" + name])
    qp = mod.QueryParams(10, beam, 1.35, 10_000_000, 10_000, mult, 10000, None, False)
    W = legs["W|" + name].astype(np.float64)
    best, reps, t_all = None, 0, time.perf_counter()
    while reps < 2 or (time.perf_counter() - t_all < args.seconds and reps < 50):
        t = time.perf_counter()
        with quiet_stdout():
            ids, dists = idx.batch_search(Q, W, args.nq, wl["method"], qp)
        dt = time.perf_counter() - t
        best = dt if best is None else min(best, dt)
        reps += 1
    gi, gd = legs["ids|" + name], legs["dists|" + name]
    same_d = (dists == gd).all(axis=1)
    same_i = (ids == gi).all(axis=1)
    # (exact scans sort unstably: rows whose distances agree may list equidistant ids in another order)
    same_set = np.array([sorted(a) == sorted(b) for a, b in zip(ids.tolist(), gi.tolist())]) & same_d
    out[name] = dict(qps=args.nq / best, reps=reps, beam=beam, mult=mult, same_dists=float(same_d.mean()), same_ids=float(same_i.mean()),
                     same_id_sets=float(same_set.mean()))
print(json.dumps(out))
