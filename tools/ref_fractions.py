"""The reference beside the GPU at EVERY window fraction of BASELINE configs[1] (SIFT-1M-like, 2-WST, optimized post-filtering):
per fraction the GPU's best setting with recall@10 > 0.95 (bench.py's sweep), then -- in one child process with PARLAY_NUM_THREADS
threads, because the reference fixes its thread count at first use -- the REAL reference (oracle/_ref) on the same graph files, the
same windows and the same setting: QPS and row-by-row comparison.  Prints one JSON object.
Usage: python tools/ref_fractions.py [--fractions=-16,...,0] [--threads 32]"""
import argparse, json, os, subprocess, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))

ap = argparse.ArgumentParser()
ap.add_argument("--fractions", default=",".join(str(p) for p in range(-16, 1)))
ap.add_argument("--threads", type=int, default=32)
ap.add_argument("--n", type=int, default=1_000_000)
ap.add_argument("--nq", type=int, default=10_000)
ap.add_argument("--seconds", type=float, default=3.0)
ap.add_argument("--cache", default="/tmp/wann_bench_cache")
ap.add_argument("--worker", default="")
args = ap.parse_args()
n, d, nq = args.n, 128, args.nq
R, L, alpha, cutoff, split = 64, 500, 1.0, 1000, 2
cache = os.path.join(args.cache, f"siftlike_n{n}_d{d}_R{R}_L{L}_c{cutoff}_s{split}") + "/"
fractions = [int(x) for x in args.fractions.split(",")]

if args.worker:  # ---- child: the real reference
    os.environ["WANN_NO_TORCH"] = "1"
    import numpy as np
    import bench
    from oracle import oracle as orc
    from util import quiet_stdout
    X, Q, labels = bench.make_data(n, d, nq, 0)
    res = np.load(args.worker)
    ref = orc.load_reference(prefer=("x86-64-v4", "native", "x86-64-v3"))
    assert ref is not None, "no reference build under oracle/_ref"
    with quiet_stdout():
        idx = ref.VamanaRangeFilterTreeIndexFloatEuclidian(X, labels, cutoff=cutoff, split_factor=split, build_params=ref.BuildParams(R, L, alpha, cache))
    out = {}
    for p in fractions:
        beam, mult = (int(x) for x in res[f"set_{p}"])
        qp = ref.QueryParams(10, beam, 1.35, 10_000_000, 10_000, mult, 10000, None, False)
        W = res[f"W_{p}"].astype(np.float64)
        best, reps, t0 = None, 0, time.perf_counter()
        while reps < 2 or (time.perf_counter() - t0 < args.seconds and reps < 20):
            t = time.perf_counter()
            with quiet_stdout():
                ids, dists = idx.batch_search(Q, W, nq, "optimized_postfilter", qp)
            dt = time.perf_counter() - t
            best = dt if best is None else min(best, dt)
            reps += 1
        gi, gd = res[f"ids_{p}"], res[f"dists_{p}"]
        same_d = (dists == gd).all(axis=1)
        same_i = (ids == gi).all(axis=1)
        same_set = np.array([sorted(a) == sorted(b) for a, b in zip(ids.tolist(), gi.tolist())])  # exact scans: equal distances may permute
        out[str(p)] = dict(qps=nq / best, reps=reps, same_dists=float(same_d.mean()), same_ids=float(same_i.mean()), same_id_sets=float((same_set & same_d).mean()))
    print(json.dumps(out))
    sys.exit(0)

import numpy as np, torch
import bench as B
import window_ann as wa
X, Q, labels = B.make_data(n, d, nq, 0)
os.makedirs(cache, exist_ok=True)
index = wa.VamanaRangeFilterTreeIndexFloatEuclidian(X, labels, cutoff=cutoff, split_factor=split, build_params=wa.BuildParams(R, L, alpha, cache))
dev = torch.device("cuda:0")
Xt = torch.from_numpy(X).to(dev); x2 = (Xt * Xt).sum(1); labt = torch.from_numpy(labels).to(dev); Qt = torch.from_numpy(Q).to(dev)
ls = np.sort(labels)
ids_t = torch.empty((nq, 10), dtype=torch.int32, device=dev); dist_t = torch.empty((nq, 10), dtype=torch.float32, device=dev)
save, gpu = {}, {}
for p in fractions:
    W = B.make_windows(ls, nq, p, 2000 + p); Wt = torch.from_numpy(W).to(dev)
    gt, gcnt = B.ground_truth(torch, Xt, x2, labt, Qt, Wt, 10)
    rows = []
    for beam, mult in B.SWEEP:
        qp = wa.QueryParams(10, beam, 1.35, 10_000_000, 10_000, mult, 10000, None, False)
        best = None
        for _ in range(3):
            t = time.perf_counter()
            index.batch_search_device(Qt.data_ptr(), Wt.data_ptr(), nq, 0, "optimized_postfilter", qp, ids_t.data_ptr(), dist_t.data_ptr(), 0)
            dt = time.perf_counter() - t
            best = dt if best is None else min(best, dt)
        rows.append((beam, mult, B.recall_of(torch, gt, gcnt, ids_t), best))
    ok = [r for r in rows if r[2] > 0.95] or [max(rows, key=lambda r: r[2])]
    beam, mult, rec, sec = min(ok, key=lambda r: r[3])
    qp = wa.QueryParams(10, beam, 1.35, 10_000_000, 10_000, mult, 10000, None, False)
    index.batch_search_device(Qt.data_ptr(), Wt.data_ptr(), nq, 0, "optimized_postfilter", qp, ids_t.data_ptr(), dist_t.data_ptr(), 0)
    save[f"W_{p}"] = W; save[f"set_{p}"] = np.array([beam, mult]); save[f"ids_{p}"] = ids_t.cpu().numpy().view(np.uint32); save[f"dists_{p}"] = dist_t.cpu().numpy()
    gpu[str(p)] = dict(beam=beam, mult=mult, recall=round(rec, 4), qps=round(nq / sec), ms=round(sec * 1e3, 3))
    print(f"[frac] 2^{p}: GPU {gpu[str(p)]}", file=sys.stderr, flush=True)
res = os.path.join(args.cache, "fractions_result.npz")
np.savez(res, **save)
del index
env = dict(os.environ, PARLAY_NUM_THREADS=str(args.threads))
pr = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", res, "--fractions=" + args.fractions, "--threads", str(args.threads), "--n", str(n),
                     "--nq", str(nq), "--seconds", str(args.seconds), "--cache", args.cache], env=env, capture_output=True, text=True, timeout=3000)
line = [x for x in pr.stdout.strip().splitlines() if x.startswith("{")]
ref = json.loads(line[-1]) if line else {"error": (pr.stderr or "")[-600:]}
out = dict(workload=f"SIFT-1M-like n={n} d={d} L2 2-WST optimized_postfilter nq={nq} k=10", reference_threads=args.threads, fractions={})
for p in fractions:
    g, r = gpu[str(p)], ref.get(str(p), {})
    out["fractions"][f"2^{p}"] = dict(gpu=g, reference=r, speedup=(round(g["qps"] / r["qps"], 1) if "qps" in r else None))
    print(f"[frac] 2^{p}: GPU {g['qps']:,} QPS vs reference {r.get('qps', 0):,.0f} QPS; rows identical (dists / ids / id sets): {r.get('same_dists')} / {r.get('same_ids')} / {r.get('same_id_sets')}", file=sys.stderr, flush=True)
if "error" in ref:
    out["error"] = ref["error"]
print(json.dumps(out))
