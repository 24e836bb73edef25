"""Dev tool: the same mid-window-fraction batches again and again on one index (n = 150 000, beams up to 5 120: companion launch,
helper-wave packets, pollers, look-aheads, evidence-first order) -- every repetition must return the rows and the work counters
of the first one (which tests/test_gpu_parity.py pins to the oracle).  Timing differs from run to run; results must not.
Usage: python tools/stress_repeat.py [seconds]"""
import os, sys, time
os.environ.setdefault("WANN_TEST_HOOKS", "1")  # this tool flips WANN_* switches between calls on one index
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import window_ann as wa
from util import sift_like, distinct_labels, windows

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
n, d, nq = 150000, 64, 1500
g = sift_like(n, d, 14); X, Q = g(n), g(nq); labels = distinct_labels(n, 16)
idx = wa.VamanaRangeFilterTreeIndexFloatEuclidian(X, labels, cutoff=1000, split_factor=2, build_params=wa.BuildParams(24, 64, 1.0, ""))
cases = [(-9, 10, 1), (-8, 80, 1), (-9, 40, 2), (-6, 40, 1), (-10, 20, 1), (-3, 64, 1), (-8, 20, 3), (-10, 10, 4), (-7, 40, 2)]
base, reps, bad = {}, 0, 0
t0 = time.time()
while time.time() - t0 < budget:
    for p, beam, mult in cases:
        W = windows(labels, nq, p, seed=5)
        ids, dists = idx.batch_search(Q, W, nq, "optimized_postfilter", wa.QueryParams(10, beam, 1.35, 10**7, 10**4, mult, 10000, None, False))
        c = idx.counters()
        cur = (ids.copy(), dists.copy(), c["beam_searches"], c["hops"], c["dist_cmps"])
        key = (p, beam, mult)
        if key not in base:
            base[key] = cur
        else:
            b = base[key]
            if not (np.array_equal(b[0], cur[0]) and np.array_equal(b[1], cur[1]) and b[2:] == cur[2:]):
                bad += 1
                print("MISMATCH", key, "rows differing", int((b[0] != cur[0]).any(1).sum()), b[2:], cur[2:], flush=True)
    reps += 1
print(f"{reps} repetitions of {len(cases)} batches in {time.time() - t0:.0f}s: {bad} mismatches; last counters {c}")
sys.exit(1 if bad else 0)
