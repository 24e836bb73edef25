"""Dev tool: device time and work counters of fixed (beam, mult) settings at given window fractions on the bench index
(SIFT-1M-like 2-WST).  Usage: python tools/frac_probe.py --fractions=-11,-9,-6 --settings 80,1:160,1 [--n 1000000]"""
import argparse, os, sys, time
os.environ.setdefault("WANN_TEST_HOOKS", "1")  # this tool flips WANN_* switches between calls on one index
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np, torch
import bench as B
import window_ann as wa

ap = argparse.ArgumentParser()
ap.add_argument("--fractions", default="-11,-9,-7,-6")
ap.add_argument("--settings", default="80,1")
ap.add_argument("--n", type=int, default=1_000_000)
ap.add_argument("--nq", type=int, default=10_000)
ap.add_argument("--reps", type=int, default=3)
args = ap.parse_args()
n, d, nq = args.n, 128, args.nq
X, Q, labels = B.make_data(n, d, nq, 0)
cache = f"/tmp/wann_bench_cache/siftlike_n{n}_d{d}_R64_L500_c1000_s2/"
os.makedirs(cache, exist_ok=True)
t0 = time.time()
index = wa.VamanaRangeFilterTreeIndexFloatEuclidian(X, labels, cutoff=1000, split_factor=2, build_params=wa.BuildParams(64, 500, 1.0, cache))
print(f"index ready in {time.time() - t0:.1f}s", flush=True)
dev = torch.device("cuda:0")
Xt = torch.from_numpy(X).to(dev); x2 = (Xt * Xt).sum(1); labt = torch.from_numpy(labels).to(dev); Qt = torch.from_numpy(Q).to(dev)
ls = np.sort(labels)
ids_t = torch.empty((nq, 10), dtype=torch.int32, device=dev); dist_t = torch.empty((nq, 10), dtype=torch.float32, device=dev)
for p in [int(x) for x in args.fractions.split(",")]:
    W = B.make_windows(ls, nq, p, 2000 + p); Wt = torch.from_numpy(W).to(dev)
    gt, gcnt = B.ground_truth(torch, Xt, x2, labt, Qt, Wt, 10)
    torch.cuda.synchronize()
    for st in args.settings.split(":"):
        beam, mult = (int(x) for x in st.split(","))
        qp = wa.QueryParams(10, beam, 1.35, 10_000_000, 10_000, mult, 10000, None, False)
        best = None
        for _ in range(args.reps):
            index.batch_search_device(Qt.data_ptr(), Wt.data_ptr(), nq, 0, "optimized_postfilter", qp, ids_t.data_ptr(), dist_t.data_ptr(), 0)
            c = index.counters()
            best = c if best is None or c["device_ms"] < best["device_ms"] else best
        rec = B.recall_of(torch, gt, gcnt, ids_t)
        print(f"2^{p} beam {beam} x{mult}: recall {rec:.4f} device {best['device_ms']:.2f} ms kernel {best['search_kernel_ms']:.2f} ms rounds {best['rounds']} "
              f"searches {best['beam_searches']} hops {best['hops']} spec_searches {best['spec_searches']} spec_hops {best['spec_hops']} handoffs {best.get('deep_handoffs', 0)} lookaheads_used {best.get('lookaheads_used', 0)} big_searches {best.get('big_searches', 0)} big_hops {best.get('big_hops', 0)} packet_hops {best.get('packet_hops', 0)} own_scorings {best.get('own_scorings', 0)} prefetched {best.get('prefetched_hops', 0)} poll_timeouts {best.get('poll_timeouts', 0)}", flush=True)
        if os.environ.get("PROBE_COMPARE"):
            ref_rows = (ids_t.clone(), dist_t.clone()); ref_c = dict(c)
            for envs in ({"WANN_NO_BIG": "1"}, {"WANN_NO_SPEC": "1"}, {"WANN_NO_POLLERS": "1"}, {}):
                os.environ.update(envs)
                c2 = None
                for _ in range(args.reps):
                    index.batch_search_device(Qt.data_ptr(), Wt.data_ptr(), nq, 0, "optimized_postfilter", qp, ids_t.data_ptr(), dist_t.data_ptr(), 0)
                    cc = index.counters()
                    c2 = cc if c2 is None or cc["device_ms"] < c2["device_ms"] else c2
                for k_ in envs: os.environ.pop(k_)
                bad = (~((ids_t == ref_rows[0]).all(1) & (dist_t == ref_rows[1]).all(1))).nonzero().flatten().tolist()
                print("   vs", envs, "rows differing:", bad[:10], "counters", {k_: (ref_c[k_], c2[k_]) for k_ in ("beam_searches", "hops", "dist_cmps") if ref_c[k_] != c2[k_]}, f"device {c2['device_ms']:.2f} ms", flush=True)
