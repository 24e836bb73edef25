import os, sys, time, numpy as np
os.environ.setdefault("WANN_TEST_HOOKS", "1")  # this tool flips WANN_* switches between calls on one index
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from util import sift_like
import window_ann as wa
n, d = 200000, 128
g = sift_like(n, d, 1234); X = g(n)
cache = "/tmp/phase_cache/"; os.makedirs(cache, exist_ok=True)
lab = np.arange(n, dtype=np.float32)
idx = wa.PostfilterVamanaIndexFloatEuclidian(X, filters=lab, build_params=wa.BuildParams(64, 500, 1.0, cache))
rows = idx.partition_graph(0, 0, 64)
BEAMS = tuple(int(x) for x in sys.argv[1].split(',')) if len(sys.argv) > 1 else (40, 80, 320)
NQS = tuple(int(x) for x in sys.argv[2].split(',')) if len(sys.argv) > 2 else (64, 8192)
for nq in NQS:
    Q = g(nq); qids = np.arange(nq, dtype=np.int64) + 10**7
    for beam in BEAMS:
        os.environ["WANN_PROFILE_PHASES"] = "1"
        ids, dists, sizes, hops, cmps = wa.raw_beam_search(0, X, rows, 0, Q, qids, beam)
        os.environ.pop("WANN_PROFILE_PHASES")
        os.environ["WANN_VERBOSE"] = "1"
        t0 = time.time(); wa.raw_beam_search(0, X, rows, 0, Q, qids, beam); t1 = time.time() - t0
        os.environ.pop("WANN_VERBOSE")
        print(f"nq={nq} beam={beam}: hops/search {hops.mean():.1f} cmps/search {cmps.mean():.0f} call {t1*1e3:.1f} ms", flush=True)
