"""Dev tool: random shared-window batches on a PrefilterIndex, dense MFMA path against the exact scan, again and again -- dimensions
up to 512, clustered and uniform data, near-duplicate points, labels with and without correlation to the geometry.  The proof
bound of k_rerank decides which queries may skip the exact scan: every row of every batch must equal the scan's.
Usage: python tools/stress_prefilter.py [seconds]"""
import os, sys, time
os.environ.setdefault("WANN_TEST_HOOKS", "1")  # this tool flips WANN_* switches between calls on one index
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import window_ann as wa

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(2026)
t0, batches, bad, unproven, dense = time.time(), 0, 0, 0, 0
qp = lambda k: wa.QueryParams(k, 10, 1.35, 10**7, 10**4, 1, 10000, None, False)
while time.time() - t0 < budget:
    d = int(rng.choice([24, 64, 100, 128, 200, 256, 384, 512]))
    n = int(rng.integers(30000, 120000))
    mips = bool(rng.integers(0, 2))
    nclu = int(rng.integers(4, 40))
    cent = rng.standard_normal((nclu, d))
    spread = float(rng.choice([0.02, 0.1, 0.3, 1.0]))
    cl = rng.integers(0, nclu, n)
    X = (cent[cl] + spread * rng.standard_normal((n, d))).astype(np.float32)
    if mips or rng.integers(0, 2): X /= np.linalg.norm(X, axis=1, keepdims=True)
    if rng.integers(0, 2): X[rng.integers(0, n, 200)] = X[rng.integers(0, n)]  # duplicates
    labels = (cl + rng.random(n)).astype(np.float32) if rng.integers(0, 2) else rng.permutation(n).astype(np.float32)
    idx = (wa.PrefilterIndexFloatMips if mips else wa.PrefilterIndexFloatEuclidian)(X, labels)
    ls = np.sort(labels)
    for _ in range(6):
        nq = int(rng.integers(200, 1500))
        qc = rng.integers(0, nclu, nq)
        Q = (cent[qc] + spread * rng.standard_normal((nq, d))).astype(np.float32)
        if rng.integers(0, 2): Q /= np.linalg.norm(Q, axis=1, keepdims=True)
        nfam = int(rng.integers(1, 12))
        W = np.zeros((nq, 2))
        fam = rng.integers(0, nfam, nq)
        for f in range(nfam):
            w = int(rng.integers(1100, min(n - 2, 40000)))
            s = int(rng.integers(0, n - w - 1))
            W[fam == f] = (ls[s] - 1e-3, ls[s + w])
        k = int(rng.choice([1, 10, 16]))
        os.environ.pop("WANN_NO_GEMM", None)
        ids, dists = idx.batch_search(Q, W, nq, qp(k))
        c = idx.counters()
        os.environ["WANN_NO_GEMM"] = "1"
        ids2, dists2 = idx.batch_search(Q, W, nq, qp(k))
        os.environ.pop("WANN_NO_GEMM", None)
        batches += 1
        dense += c["gemm_queries"]
        unproven += c["gemm_unproven"]
        if not (np.array_equal(dists, dists2) and np.array_equal(ids, ids2)):
            bad += 1
            print("MISMATCH", dict(d=d, n=n, mips=mips, nq=nq, k=k, rows=int(((dists != dists2) | (ids != ids2)).any(1).sum())), c, flush=True)
print(f"{batches} batches in {time.time() - t0:.0f}s: {dense} queries on the dense path, {unproven} unproven, {bad} batches with a mismatch")
sys.exit(1 if bad else 0)
