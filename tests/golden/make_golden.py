"""Generate the golden fixtures in this directory from the REAL reference.

Runs only in the authoring container: it imports the reference's own pybind11 module built from
/root/reference by `make -C oracle ref REF_MARCH=native` (oracle/_ref/native/, g++ 11.4 -O3
-march=native on a Cooperlake AVX-512/FMA host -- fp32 evaluation order is compiler specific, see
SURVEY.md A.3).  Only DATA is written here: inputs, the graph cache files the reference built, and
the (ids, dists) it returned.  Usage:  python tests/golden/make_golden.py
"""
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from util import REPO, distinct_labels, quiet_stdout, sift_like, unit_mixture, windows  # noqa: E402

sys.path.insert(0, REPO)
from oracle import oracle as orc  # noqa: E402

K = 10
BEAMS_MULTS = [(10, 1), (40, 1), (40, 2), (10, 4)]
FRACTIONS = [-7, -5, -3, -1, 0]
TREE_METHODS = ["optimized_postfilter", "fenwick", "three_split"]


def qp(ref, beam, mult, max_beam=10000, ratio=None):
    return ref.QueryParams(K, beam, 1.35, 10_000_000, 10_000, mult, max_beam, ratio, False)


def make(name, metric, X, Q, labels, R, L, cutoff, elem="Float", alpha=1.0):
    ref = orc.load_reference(prefer=("native",))
    assert ref is not None, "build the reference first: make -C oracle ref REF_MARCH=native"
    sfx = elem + ("Mips" if metric == "mips" else "Euclidian")
    nq = Q.shape[0]
    tmp = tempfile.mkdtemp(prefix="golden_")
    out = {"X": X, "Q": Q, "labels": labels, "meta": np.array([R, L, cutoff, K, int(round(alpha * 1000))], dtype=np.int64)}
    W = {p: windows(labels, nq, p, seed=100 + p) for p in FRACTIONS}
    # one window entirely outside the label span and one of width zero, in a separate batch
    Wedge = np.array([[5.0, 6.0], [-3.0, -2.0], [labels[7], labels[7]], [0.25, 0.2501]] * (nq // 4), dtype=np.float64)
    for p in FRACTIONS:
        out[f"W_{p}"] = W[p]
    out["W_edge"] = Wedge
    kinds = {
        "VamanaRangeFilterTreeIndex": dict(cutoff=cutoff, split_factor=2),
        "SuperOptimizedPostfilterTreeIndex": dict(cutoff=cutoff, split_factor=2, shift_factor=0.5),
        "PostfilterVamanaIndex": dict(),
        "RangeFilterTreeIndex": dict(cutoff=cutoff, split_factor=2),
        "PrefilterIndex": dict(),
    }
    files = {}
    for kind, kw in kinds.items():
        cdir = os.path.join(tmp, kind) + "/"
        os.makedirs(cdir)
        labkw = "filters" if kind == "PostfilterVamanaIndex" else "filter_values"
        with quiet_stdout():
            idx = getattr(ref, kind + sfx)(X, **{labkw: labels}, build_params=ref.BuildParams(R, L, alpha, cdir), **kw)
        for f in sorted(os.listdir(cdir)):
            files[f"{kind}/{f}"] = np.frombuffer(open(cdir + f, "rb").read(), dtype=np.uint8)
        methods = TREE_METHODS if kind.endswith("RangeFilterTreeIndex") else [""]
        for method in methods:
            for beam, mult in BEAMS_MULTS:
                for p in FRACTIONS + ["edge"]:
                    if kind == "PrefilterIndex" and (p == "edge" or p < -5):
                        continue  # fewer than k points in the window: the reference reads past its result (UB)
                    if kind == "PrefilterIndex" and (beam, mult) != BEAMS_MULTS[0]:
                        continue  # beam parameters are irrelevant to brute force
                    Wp = Wedge if p == "edge" else W[p]
                    args = (Q, Wp, nq) + ((method,) if method else ())
                    with quiet_stdout():
                        ids, dists = idx.batch_search(*args, qp(ref, beam, mult))
                    key = f"{kind}|{method}|{beam}|{mult}|{p}"
                    out["ids|" + key] = ids
                    out["dists|" + key] = dists
        if kind == "VamanaRangeFilterTreeIndex":  # beam >= max_beam -> empty; ratio fallback
            with quiet_stdout():
                ids, dists = idx.batch_search(Q, W[-3], nq, "optimized_postfilter", qp(ref, 64, 1, max_beam=64))
            out["ids|VamanaRangeFilterTreeIndex|maxbeam"] = ids
            out["dists|VamanaRangeFilterTreeIndex|maxbeam"] = dists
            with quiet_stdout():
                ids, dists = idx.batch_search(Q, W[-3], nq, "optimized_postfilter", qp(ref, 8, 1, max_beam=20))
            out["ids|VamanaRangeFilterTreeIndex|overshoot"] = ids
            out["dists|VamanaRangeFilterTreeIndex|overshoot"] = dists
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)
    np.savez_compressed(os.path.join(HERE, f"{name}_graphs.npz"), **files)
    shutil.rmtree(tmp)
    print(name, "arrays", len(out), "graph files", len(files))


if __name__ == "__main__":
    ONLY = set(sys.argv[1:])
    _make = make

    def make(name, *a, **k):  # noqa: F811  (python make_golden.py u8_l2 i8_mips: regenerate only these)
        if not ONLY or name in ONLY:
            _make(name, *a, **k)
    n, d, nq = 2048, 32, 48
    g = sift_like(n, d, 1234)
    make("sift_l2", "Euclidian", g(n), g(nq), distinct_labels(n, 7), R=16, L=32, cutoff=200)
    n, d = 1536, 100
    g = unit_mixture(n, d, 99)
    make("unit_mips", "mips", g(n), g(nq), distinct_labels(n, 8), R=16, L=32, cutoff=200)
    # byte variants (python_bindings.cpp:234-237; int32 distances cast to float, euclidian_point.h:44-60, mips_point.h:44-58)
    n, d, nq = 1024, 24, 32
    g = sift_like(n, d, 4321)
    make("u8_l2", "Euclidian", g(n).astype(np.uint8), g(nq).astype(np.uint8), distinct_labels(n, 9), R=16, L=32, cutoff=200, elem="UInt8")
    g = sift_like(n, d, 8765)
    make("i8_mips", "mips", (g(n) - 128).astype(np.int8), (g(nq) - 128).astype(np.int8), distinct_labels(n, 10), R=16, L=32, cutoff=200, elem="Int8")
    # 512 bytes per row: sums beyond 2^24, exact only with the reference's int32 accumulation
    n, d, nq = 640, 512, 16
    g = sift_like(n, d, 2468)
    make("u8_l2_d512", "Euclidian", g(n).astype(np.uint8), g(nq).astype(np.uint8), distinct_labels(n, 11), R=16, L=32, cutoff=200, elem="UInt8")
    # max_degree 96 (graph.h:115-124 takes any R): with alpha = 1.35 a quarter of the rows of the large partitions list more than
    # 64 neighbours -- the kernels work such rows in two halves
    n, d, nq = 3000, 32, 48
    g = sift_like(n, d, 777)
    make("sift_l2_r96", "Euclidian", g(n), g(nq), distinct_labels(n, 21), R=96, L=192, cutoff=400, alpha=1.35)
