"""BASELINE.json's full-size configurations as data: the laws, seeds, classes and constructor arguments that bench.py,
tools/bench_configs.py and tools/bench_prefilter.py use, in one place for tests/test_gpu_fullsize.py and its reference child
(tools/ref_rows.py).  `mod` is the module that provides the classes: the product (`window_ann`) or the real reference."""
import os

import numpy as np

K = 10
R, L, ALPHA = 64, 500, 1.0

CONFIGS = {
    # configs[1]: SIFT-1M-like, 2-ary window search tree, optimized post-filtering (the headline workload)
    "sift": dict(n=1_000_000, d=128, nq=10_000, cls="VamanaRangeFilterTreeIndexFloatEuclidian", kw=dict(cutoff=1000, split_factor=2),
                 method="optimized_postfilter", graphs=True),
    # configs[2]: GloVe-1.18M-like (unit-norm rows, inner product), super-optimised post-filter tree
    "glove": dict(n=1_183_514, d=100, nq=10_000, cls="SuperOptimizedPostfilterTreeIndexFloatMips", kw=dict(cutoff=1000, split_factor=2, shift_factor=0.5),
                  method=None, graphs=True),
    # configs[3]: deep-10M-like, 4-ary tree (on ONE GPU here)
    "deep": dict(n=9_990_000, d=96, nq=10_000, cls="VamanaRangeFilterTreeIndexFloatMips", kw=dict(cutoff=1000, split_factor=4),
                 method="optimized_postfilter", graphs=True),
    # configs[3] as BASELINE.json's text states it ("96-d L2"; the reference's own deep runs are inner product): the same rows and
    # tree under squared L2 -- at full size (opt-in: two more minutes of build) and at n = 10^6 (always)
    "deep_l2": dict(n=9_990_000, d=96, nq=10_000, cls="VamanaRangeFilterTreeIndexFloatEuclidian", kw=dict(cutoff=1000, split_factor=4),
                    method="optimized_postfilter", graphs=True),
    "deep_l2_1m": dict(n=1_000_000, d=96, nq=10_000, cls="VamanaRangeFilterTreeIndexFloatEuclidian", kw=dict(cutoff=1000, split_factor=4),
                       method="optimized_postfilter", graphs=True),
    # configs[4]: adversarial clusters, stand-alone PrefilterIndex (dense MFMA path on native windows, exact scan on 2^-12 windows)
    "adverse": dict(n=1_000_000, d=100, nq=9_900, cls="PrefilterIndexFloatMips", kw={}, method=None, graphs=False),
}


def make_data(name):
    import bench
    if name == "sift":
        return bench.make_data(1_000_000, 128, 10_000, 1)
    if name in ("glove", "deep", "deep_l2", "deep_l2_1m"):
        from util import unit_mixture
        cfg = CONFIGS[name]
        g = unit_mixture(cfg["n"], cfg["d"], 2025)
        X, Q = g(cfg["n"]), g(cfg["nq"])
        labels = ((np.random.default_rng(77).permutation(cfg["n"]) + 0.5) / cfg["n"]).astype(np.float32)
        return X, Q, labels
    # adversarial data (experiments/generate_advserial_dataset.py:8-69), the law of tools/bench_prefilter.py
    rng = np.random.default_rng(0)
    nclu, per, d = 100, 10000, 100
    n = nclu * per
    cent = rng.standard_normal((nclu, d)).astype(np.float32)
    X = cent[np.repeat(np.arange(nclu), per)] + 0.1 * rng.standard_normal((n, d)).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    labels = (np.repeat(np.arange(nclu), per) - 0.5 + rng.random(n)).astype(np.float32)
    qc = np.repeat(np.arange(nclu), 99)
    Q = cent[(qc + 1 + rng.integers(0, nclu - 1, qc.size)) % nclu] + 0.1 * rng.standard_normal((qc.size, d)).astype(np.float32)
    Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    return X, Q.astype(np.float32), labels


def native_windows():
    """adverse: the dataset's own windows, one cluster each"""
    qc = np.repeat(np.arange(100), 99)
    return np.stack([qc - 0.5, qc + 0.5], 1).astype(np.float32)


def fraction_windows(labels, nq, p, seed):
    """nq windows that hold a 2^p fraction of the points each (bench.make_windows)"""
    import bench
    return bench.make_windows(np.sort(labels), nq, p, seed)


def make_index(mod, name, X, labels, cache):
    cfg = CONFIGS[name]
    if not cfg["graphs"]:
        return getattr(mod, cfg["cls"])(X, labels)
    os.makedirs(cache, exist_ok=True)
    return getattr(mod, cfg["cls"])(X, labels, build_params=mod.BuildParams(R, L, ALPHA, cache), **cfg["kw"])


def query_params(mod, beam, mult):
    return mod.QueryParams(K, beam, 1.35, 10_000_000, 10_000, mult, 10000, None, False)
