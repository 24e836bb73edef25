"""Dev tool: per-phase cycles of the register-resident small-beam core (wave_beam_search_small: beams <= 128) under load, on a
stand-alone graph -- squared L2 (d = 128, SIFT-like) beside inner product (d = 96 / 100, unit-norm mixture rows), to see where
the hop of the MIPS legs goes.  Needs a `make PROFILE=1` build on LD_LIBRARY_PATH for the phase lines.
Usage: python tools/phase_profile_small.py [n] [beams] [nqs] [metric:d]"""
import os, sys, time, numpy as np
os.environ.setdefault("WANN_TEST_HOOKS", "1")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from util import sift_like, unit_mixture
import window_ann as wa
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
BEAMS = tuple(int(x) for x in sys.argv[2].split(',')) if len(sys.argv) > 2 else (40, 80)
NQS = tuple(int(x) for x in sys.argv[3].split(',')) if len(sys.argv) > 3 else (10000,)
ONLY = sys.argv[4] if len(sys.argv) > 4 else ""  # e.g. "1:96" = inner product, d = 96 only
for metric, d, gen, cls in ((0, 128, sift_like, "PostfilterVamanaIndexFloatEuclidian"), (1, 96, unit_mixture, "PostfilterVamanaIndexFloatMips"),
                            (1, 100, unit_mixture, "PostfilterVamanaIndexFloatMips")):
    if ONLY and ONLY != f"{metric}:{d}":
        continue
    g = gen(n, d, 1234); X = g(n)
    cache = f"/tmp/phase_cache_{metric}_{d}/"; os.makedirs(cache, exist_ok=True)
    lab = np.arange(n, dtype=np.float32)
    t0 = time.time()
    idx = getattr(wa, cls)(X, filters=lab, build_params=wa.BuildParams(64, 500, 1.0, cache))
    rows = idx.partition_graph(0, 0, 64)
    print(f"== metric {metric} d {d} n {n}: graph in {time.time() - t0:.0f}s", flush=True)
    del idx
    for nq in NQS:
        Q = g(nq); qids = np.arange(nq, dtype=np.int64) + 10**7
        for beam in BEAMS:
            os.environ["WANN_PROFILE_PHASES"] = "1"
            ids, dists, sizes, hops, cmps = wa.raw_beam_search(metric, X, rows, 0, Q, qids, beam)
            os.environ.pop("WANN_PROFILE_PHASES")
            os.environ["WANN_VERBOSE"] = "1"
            for _ in range(3):
                wa.raw_beam_search(metric, X, rows, 0, Q, qids, beam)
            os.environ.pop("WANN_VERBOSE")
            print(f"nq={nq} beam={beam}: hops/search {hops.mean():.1f} cmps/search {cmps.mean():.0f}; bytes/search {cmps.mean() * 4 * ((d + 15) // 16 * 16) + hops.mean() * 256:.0f}", flush=True)
