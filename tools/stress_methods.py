"""Dev tool: fenwick / three_split / optimized_postfilter batches interleaved on one tree index, again and again -- the end scans of
the multi-task methods run on a stream of their own beside the graph searches (round 6), so timing differs from run to run; rows
and work counters must not: every repetition must return those of the first one (which tests/test_gpu_parity.py pins to the
oracle at smaller sizes), through the blocking call and through the asynchronous lanes.  Usage: python tools/stress_methods.py [seconds]"""
import os, sys, time
os.environ.setdefault("WANN_TEST_HOOKS", "1")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np, torch
import window_ann as wa
from util import unit_mixture, distinct_labels, windows

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
n, d, nq, k = 200000, 64, 2000, 10
g = unit_mixture(n, d, 21); X, Q = g(n), g(nq); labels = distinct_labels(n, 22)
idx = wa.VamanaRangeFilterTreeIndexFloatMips(X, labels, cutoff=1000, split_factor=2, build_params=wa.BuildParams(32, 64, 1.0, ""))
dev = torch.device("cuda:0")
Qt = torch.from_numpy(Q).to(dev)
cases = [("fenwick", -6, 10, 1), ("three_split", -6, 20, 1), ("optimized_postfilter", -6, 20, 1), ("fenwick", -4, 10, 2), ("three_split", -9, 10, 1),
         ("fenwick", -12, 10, 1), ("optimized_postfilter", -12, 10, 1), ("three_split", -3, 20, 2)]
Ws = {p: windows(labels, nq, p, seed=3).astype(np.float32) for p in {c[1] for c in cases}}
Wt = {p: torch.from_numpy(w).to(dev) for p, w in Ws.items()}
outs = [(torch.empty((nq, k), dtype=torch.int32, device=dev), torch.empty((nq, k), dtype=torch.float32, device=dev)) for _ in range(3)]
base, reps, bad = {}, 0, 0
t0 = time.time()
while time.time() - t0 < budget:
    tickets = []
    for i, (m, p, beam, mult) in enumerate(cases):
        qp = wa.QueryParams(k, beam, 1.35, 10**7, 10**4, mult, 10000, None, False)
        if reps % 2 == 0:  # blocking host-buffer call
            ids, dists = idx.batch_search(Q, Ws[p], nq, m, qp)
            c = idx.counters()
            cur = (ids.view(np.uint32).copy(), dists.copy(), c["beam_searches"], c["hops"], c["dist_cmps"], c["brute_rows"])
        else:              # asynchronous lanes, two in flight
            oi, od = outs[i % 3]
            t = idx.batch_search_device_async(Qt.data_ptr(), Wt[p].data_ptr(), nq, 0, m, qp, oi.data_ptr(), od.data_ptr(), 0)
            c = idx.wait(t)  # (wait at once: the next submission overlaps this batch's tail only through the other lane)
            cur = (oi.cpu().numpy().view(np.uint32).copy(), od.cpu().numpy().copy(), c["beam_searches"], c["hops"], c["dist_cmps"], c["brute_rows"])
        key = (m, p, beam, mult)
        if key not in base:
            base[key] = cur
        else:
            b = base[key]
            if not (np.array_equal(b[0], cur[0]) and np.array_equal(b[1], cur[1]) and b[2:] == cur[2:]):
                bad += 1
                print("MISMATCH", key, "rows differing", int((b[0] != cur[0]).any(1).sum()), b[2:], cur[2:], flush=True)
    reps += 1
print(f"{reps} repetitions of {len(cases)} batches in {time.time() - t0:.0f}s: {bad} mismatches")
sys.exit(1 if bad else 0)
