"""Shared helpers for the test-suite: seeded synthetic inputs shaped like the reference's datasets
 and stdout silencing for the chatty reference module."""
from __future__ import annotations

import contextlib
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


@contextlib.contextmanager
def quiet_stdout():
    """Silence C++ std::cout of the reference module (build timers, cache messages)."""
    sys.stdout.flush()
    saved = os.dup(1)
    devnull = os.open(os.devnull, os.O_WRONLY)
    os.dup2(devnull, 1)
    try:
        yield
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)
        os.close(devnull)


def sift_like(n, d, seed, latent=16):
    """Integer-valued floats in [0,255] with low intrinsic dimension: fp32 sums are exact in any order."""
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((latent, d))

    def gen(m):
        z = rng.standard_normal((m, latent))
        x = z @ A * 18 + 128 + rng.standard_normal((m, d)) * 6
        return np.clip(np.rint(x), 0, 255).astype(np.float32)

    return gen


def unit_mixture(n, d, seed, latent=24, clusters=50):
    """Unit-norm rows of a Gaussian mixture (GloVe/deep-like, used with MIPS)."""
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((latent, d)) / np.sqrt(latent)
    cent = rng.standard_normal((clusters, latent)) * 1.5

    def gen(m):
        c = rng.integers(0, clusters, m)
        z = cent[c] + rng.standard_normal((m, latent))
        x = z @ A + 0.1 * rng.standard_normal((m, d))
        x /= np.linalg.norm(x, axis=1, keepdims=True)
        return x.astype(np.float32)

    return gen


def distinct_labels(n, seed):
    rng = np.random.default_rng(seed)
    return ((rng.permutation(n) + 0.5) / n).astype(np.float32)


def windows(labels, nq, p, seed):
    """nq label windows covering a 2^p fraction of the points (generate_datasets/filter_generation_utils.py:9-74,
    simplified: start uniform, [s[start], s[start+w]]; p = 0 -> everything)."""
    rng = np.random.default_rng(seed)
    n = len(labels)
    s = np.sort(labels)
    w = max(1, int(n * 2.0 ** p))
    out = np.zeros((nq, 2), dtype=np.float64)
    for i in range(nq):
        if w >= n - 2:
            out[i] = (s[0] - 1, s[-1] + 1)
        else:
            st = int(rng.integers(1, n - w - 1))
            out[i] = (s[st], s[st + w])
    return out


def brute_force_gt(X, labels, Q, W, k, metric):
    """Exact filtered top-k ids (float64 arithmetic)."""
    out = []
    Xd = X.astype(np.float64)
    for q, (lo, hi) in zip(Q.astype(np.float64), W):
        idx = np.nonzero((labels >= np.float32(lo)) & (labels <= np.float32(hi)))[0]
        if metric == "mips":
            dist = -(Xd[idx] @ q)
        else:
            dist = ((Xd[idx] - q) ** 2).sum(axis=1)
        out.append(idx[np.argsort(dist, kind="stable")[:k]])
    return out


def recall(gt, ids, k=10):
    """mean |gt ∩ res[:k]| / |gt|  (experiments/run_our_method.py:174-180)"""
    tot = 0.0
    cnt = 0
    for g, r in zip(gt, ids):
        if len(g) == 0:
            continue
        tot += len(set(g.tolist()) & set(r[:k].tolist())) / len(g)
        cnt += 1
    return tot / max(cnt, 1)
