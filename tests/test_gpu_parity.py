"""GPU: the HIP path (through the pybind11 shim -> C ABI -> gfx950 kernels) against the oracle on
the same seeded inputs and against the golden vectors of the real reference.  Integer / index
results must be bit-exact; float distances are compared bit-exactly too (the kernels reproduce the
reference's fp32 evaluation order), which is stricter than north_star's 1e-5 relative bound."""
import os

import numpy as np
import pytest

import golden_util as gu
from util import REPO, distinct_labels, sift_like, unit_mixture, windows

pytestmark = pytest.mark.gpu
FLT_MAX = np.finfo(np.float32).max


def _qp(mod, beam, mult=1, k=10, max_beam=10000, ratio=None):
    return mod.QueryParams(k, beam, 1.35, 10_000_000, 10_000, mult, max_beam, ratio, False)


# ------------------------------------------------------------------------------------------
# raw kernel: one beam search per query on one graph
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("metric,gen,d", [(0, sift_like, 128), (1, unit_mixture, 100), (0, unit_mixture, 40),
                                          (1, unit_mixture, 512), (0, unit_mixture, 200)])  # (d = 512 MIPS: the RedCaps shape)
@pytest.mark.parametrize("beam", [1, 7, 10, 40, 64, 65, 100, 160, 700, 2500])
def test_raw_beam_search_matches_oracle(oracle, wa, gpu, metric, gen, d, beam):
    n, nq, R, L = 3000, 96, 32, 64
    g = gen(n, d, 11)
    X, Q = g(n), g(nq)
    Xp = oracle.pad_rows(X)
    start, sn = 500, 2000  # a partition in the middle of the point set
    rows = oracle.vamana_build(Xp, d, metric, start, sn, R, L, 1.0)
    qids = np.arange(nq, dtype=np.int64)
    qids[::3] += 700  # some queries carry an id that names a node of the partition (self-skip quirk)
    ids, dists, sizes, hops, cmps = wa.raw_beam_search(metric, X, rows, start, Q, qids, beam)
    for i in range(nq):
        oi, od, vi, vd, dc = oracle.beam_search(rows, Xp, d, metric, start, Q[i], int(qids[i]), beam)
        m = int(sizes[i])
        assert m == len(oi), (i, m, len(oi))
        assert np.array_equal(ids[i, :m], oi), (i, ids[i, :m], oi)
        assert np.array_equal(dists[i, :m], od), i
        assert int(hops[i]) == len(vi), (i, hops[i], len(vi))
        assert int(cmps[i]) == dc, (i, cmps[i], dc)


def _hash64_2(x):
    """parlay::hash64_2 (parlay/utilities.h:145-150)"""
    m = (1 << 64) - 1
    x &= m
    x = ((x ^ (x >> 30)) * 0xbf58476d1ce4e5b9) & m
    x = ((x ^ (x >> 27)) * 0x94d049bb133111eb) & m
    return x ^ (x >> 31)


@pytest.mark.parametrize("env", [{}, {"WANN_FORCE_GENERAL": "1"}, {"WANN_OLD_GENERAL": "1"}, {"WANN_RAW_BIG_LDS": "1"},
                                 {"WANN_RAW_BIG_LDS": "1", "WANN_FORCE_GENERAL": "1"},
                                 {"WANN_RAW_BIG_LDS": "1", "WANN_FORCE_GENERAL": "1", "WANN_NO_HELPER": "1"}],  # (the search wave without its helper waves)
                         ids=lambda e: "+".join(sorted(e)) or "default")
@pytest.mark.parametrize("metric,gen,d", [(0, sift_like, 128), (1, unit_mixture, 100)])
def test_raw_beam_search_core_variants(oracle, wa, gpu, monkeypatch, env, metric, gen, d):
    """Every beam-search core (register-resident, second-generation general with the exact seen set / delta list / tagged
    filter -- fed by its scoring helper waves and without them --, first-generation general, one- and four-wave kernels)
    against the oracle: ids, distances, hops, dist_cmps.
    The graph has rows that list the start node twice with a node of the same filter slot in between -- the one case in
    which the reference's multiset union keeps two copies of an entry."""
    n, nq, R, L = 6000, 64, 32, 64
    g = gen(n, d, 12)
    X, Q = g(n), g(nq)
    Xp = oracle.pad_rows(X)
    start, sn = 300, 5000
    rows = oracle.vamana_build(Xp, d, metric, start, sn, R, L, 1.0).copy()
    # (0, b, 0) with hash(b) = hash(0) in the 2^10-slot filter of the small beams: ~5 such b among 5000 nodes
    same = [b for b in range(1, sn) if (_hash64_2(b) ^ _hash64_2(0)) & 1023 == 0]
    assert same
    rng = np.random.default_rng(3)
    for r in rng.choice(sn, 300, replace=False):
        b = same[int(r) % len(same)]
        deg = max(int(rows[r, 0]), 3)
        rows[r, 0] = deg
        rows[r, 1:4] = (0, b, 0)
    for k_, v in env.items():
        monkeypatch.setenv(k_, v)
    qids = np.arange(nq, dtype=np.int64) + 10**6
    for beam in (16, 40, 100, 160, 300, 1000, 2500):
        ids, dists, sizes, hops, cmps = wa.raw_beam_search(metric, X, rows, start, Q, qids, beam)
        for i in range(nq):
            oi, od, vi, vd, dc = oracle.beam_search(rows, Xp, d, metric, start, Q[i], int(qids[i]), beam)
            m = int(sizes[i])
            assert m == len(oi), (beam, i, m, len(oi))
            assert np.array_equal(ids[i, :m], oi), (beam, i)
            assert np.array_equal(dists[i, :m], od), (beam, i)
            assert int(hops[i]) == len(vi) and int(cmps[i]) == dc, (beam, i, hops[i], len(vi), cmps[i], dc)


@pytest.mark.parametrize("env", [{}, {"WANN_FORCE_GENERAL": "1"}, {"WANN_RAW_BIG_LDS": "1", "WANN_FORCE_GENERAL": "1"}],
                         ids=lambda e: "+".join(sorted(e)) or "default")
@pytest.mark.parametrize("metric", [0, 1])
def test_raw_beam_search_on_tie_heavy_data(oracle, wa, gpu, monkeypatch, env, metric):
    """Small integer coordinates (d = 6, values below 12) and duplicate rows: most distances are shared by many nodes, so a candidate
    whose distance EQUALS the beam's last entry's is the common case.  The reference rejects it (beamSearch.h:135-145: dist >= cutoff);
    a core that tests candidates against a cutoff that is not the exact one admits it, and with a smaller id it then displaces the
    B-th entry (the third-generation core's deferred union).  Ids, distances, hops and dist_cmps against the oracle at
    beams on both sides of every core boundary."""
    n, nq, R, L, d = 4000, 48, 32, 64, 6
    rng = np.random.default_rng(99)
    X = rng.integers(0, 12, size=(n, d)).astype(np.float32)
    X[rng.choice(n, 600, replace=False)] = X[rng.choice(n, 600)]  # duplicate rows
    Q = rng.integers(0, 12, size=(nq, d)).astype(np.float32)
    Q[::4] = X[rng.choice(n, len(Q[::4]))]                        # queries that ARE points
    Xp = oracle.pad_rows(X)
    start, sn = 200, 3600
    rows = oracle.vamana_build(Xp, d, metric, start, sn, R, L, 1.0)
    for k_, v in env.items():
        monkeypatch.setenv(k_, v)
    qids = np.arange(nq, dtype=np.int64) + 10**6
    bad = []
    for beam in (20, 64, 100, 130, 160, 250, 320, 640, 1000, 1280, 2000):
        ids, dists, sizes, hops, cmps = wa.raw_beam_search(metric, X, rows, start, Q, qids, beam)
        for i in range(nq):
            oi, od, vi, vd, dc = oracle.beam_search(rows, Xp, d, metric, start, Q[i], int(qids[i]), beam)
            m = int(sizes[i])
            if not (m == len(oi) and np.array_equal(ids[i, :m], oi) and np.array_equal(dists[i, :m], od)
                    and int(hops[i]) == len(vi) and int(cmps[i]) == dc):
                bad.append((beam, i, m, len(oi), int(hops[i]), len(vi), int(cmps[i]), dc))
    assert not bad, (len(bad), bad[:10])


@pytest.mark.parametrize("metric,gen,d,R", [(0, sift_like, 64, 96), (1, unit_mixture, 100, 128), (0, unit_mixture, 40, 80)])
def test_raw_beam_search_on_wide_rows(oracle, wa, gpu, metric, gen, d, R):
    """64 < max_degree <= 128 (graph.h:115-124 takes any R): rows of up to 128 neighbours, worked in two halves per hop, against
    the oracle -- ids, distances, hops, dist_cmps at small and large beams, degree limits that cut inside the second half, and
    rows that list a node in BOTH halves with a node of the same filter slot in between (the reference's multiset union then
    keeps two copies: the second half's union counts the first half's candidates)."""
    n, nq, L = 4000, 48, 2 * R
    g = gen(n, d, 31)
    X, Q = g(n), g(nq)
    Xp = oracle.pad_rows(X)
    start, sn = 200, 3500
    rows = oracle.vamana_build(Xp, d, metric, start, sn, R, L, 1.35).copy()
    rng = np.random.default_rng(5)
    # (inner-product builds prune hard: a third of the rows get further, random neighbours up to 65 .. R -- the search does not
    # care where a row came from)
    for r in rng.choice(sn, sn // 3, replace=False):
        have = set(int(x) for x in rows[r, 1:1 + rows[r, 0]])
        want = int(rng.integers(65, R + 1))
        while len(have) < want:
            have.add(int(rng.integers(0, sn)))
        lst = list(rows[r, 1:1 + rows[r, 0]]) + sorted(have - set(int(x) for x in rows[r, 1:1 + rows[r, 0]]))
        rows[r, 0] = len(lst)
        rows[r, 1:1 + len(lst)] = lst
    deg = rows[:, 0]
    assert (deg > 64).sum() > 500 and deg.max() <= R
    # rows (.., 0 at slot 3, .., b, 0 at slots 70, 71) with hash(b) = hash(0) in the 2^10-slot filter: node 0 passes the filter twice
    same = [b for b in range(1, sn) if (_hash64_2(b) ^ _hash64_2(0)) & 1023 == 0]
    wide = np.flatnonzero(deg > 72)
    for r in rng.choice(wide, min(200, len(wide)), replace=False):
        rows[r, 1 + 3] = 0
        rows[r, 1 + 70] = same[int(r) % len(same)]
        rows[r, 1 + 71] = 0
    qids = np.arange(nq, dtype=np.int64) + 10**6
    qids[::4] = np.arange(0, nq, 4) + 100  # (some queries carry the id of a node: the self-skip quirk)
    for beam, limit, dl in ((10, 10**7, 10**4), (40, 10**7, 10**4), (100, 10**7, 10**4), (160, 10**7, 10**4), (700, 10**7, 10**4),
                            (2500, 10**7, 10**4), (40, 10**7, 70), (100, 25, 100)):
        ids, dists, sizes, hops, cmps = wa.raw_beam_search(metric, X, rows, start, Q, qids, beam, limit, dl)
        for i in range(nq):
            oi, od, vi, vd, dc = oracle.beam_search(rows, Xp, d, metric, start, Q[i], int(qids[i]), beam, limit=limit, degree_limit=dl)
            m = int(sizes[i])
            assert m == len(oi), (beam, i, m, len(oi))
            assert np.array_equal(ids[i, :m], oi), (beam, dl, i)
            assert np.array_equal(dists[i, :m], od), (beam, i)
            assert int(hops[i]) == len(vi) and int(cmps[i]) == dc, (beam, i, hops[i], len(vi), cmps[i], dc)


def test_wide_row_index_matches_oracle(oracle, wa, gpu, tmp_path):
    """an index with max_degree 96 end to end: built by the product (the host builder: byte-identical graph files to the oracle's),
    searched by the two-halves core in the one-wave kernel -- rows and work counters equal the oracle's; R > 128 is refused"""
    n, d, nq = 5000, 48, 300
    g = sift_like(n, d, 61)
    X, Q = g(n), g(nq)
    labels = distinct_labels(n, 23)
    c1, c2 = str(tmp_path / "a") + "/", str(tmp_path / "b") + "/"
    os.makedirs(c1), os.makedirs(c2)
    idx = wa.VamanaRangeFilterTreeIndexFloatEuclidian(X, labels, cutoff=600, split_factor=2, build_params=wa.BuildParams(96, 192, 1.35, c1))
    oi = oracle.VamanaRangeFilterTreeIndexFloatEuclidian(X, labels, cutoff=600, split_factor=2, build_params=oracle.BuildParams(96, 192, 1.35, c2))
    for f in sorted(os.listdir(c2)):
        assert open(c1 + f, "rb").read() == open(c2 + f, "rb").read(), f
    assert idx.max_degree() == 96
    for p, method, beam, mult in ((-2, "optimized_postfilter", 20, 2), (-5, "optimized_postfilter", 10, 1), (-7, "optimized_postfilter", 10, 2),
                                  (-4, "fenwick", 20, 1), (-3, "three_split", 20, 2)):
        W = windows(labels, nq, p, 70 + p)
        ids, dists = idx.batch_search(Q, W.astype(np.float32), nq, method, _qp(wa, beam, mult))
        eids, edists = oi.batch_search(Q, W, nq, method, _qp(oracle, beam, mult))
        ok, why = gu.same_rows(eids, edists, ids, dists, method != "optimized_postfilter", gu.RowContext(X, labels, Q, W, "l2"))
        assert ok, (p, method, why)
        c = idx.counters()
        assert (c["beam_searches"], c["hops"]) == (oi.last_counters["searches"], oi.last_counters["hops"]), (p, method)
        assert c["dist_cmps"] + c["brute_rows"] == oi.last_counters["dist_cmps"], (p, method)  # (the oracle counts scanned rows there too)
    with pytest.raises(RuntimeError):
        wa.VamanaRangeFilterTreeIndexFloatEuclidian(X, labels, cutoff=600, split_factor=2, build_params=wa.BuildParams(129, 192, 1.0, ""))


def test_raw_beam_search_limits(oracle, wa, gpu):
    n, d, nq = 1500, 32, 40
    g = sift_like(n, d, 5)
    X, Q = g(n), g(nq)
    Xp = oracle.pad_rows(X)
    rows = oracle.vamana_build(Xp, d, 0, 0, n, 24, 48, 1.0)
    qids = np.arange(nq, dtype=np.int64) + 10**6
    for limit, dl in [(5, 10000), (10**7, 7), (3, 3)]:
        ids, dists, sizes, hops, cmps = wa.raw_beam_search(0, X, rows, 0, Q, qids, 20, limit, dl)
        for i in range(nq):
            oi, od, vi, vd, dc = oracle.beam_search(rows, Xp, d, 0, 0, Q[i], int(qids[i]), 20, limit=limit, degree_limit=dl)
            assert np.array_equal(ids[i, :sizes[i]], oi) and np.array_equal(dists[i, :sizes[i]], od)
            assert int(hops[i]) == len(vi) and int(cmps[i]) == dc


# ------------------------------------------------------------------------------------------
# golden vectors of the real reference, through reference-built graph files
# ------------------------------------------------------------------------------------------
SUPPORTED = lambda kind, method: True  # noqa: E731  (every query method runs on the device)


@pytest.mark.parametrize("name", list(gu.FIXTURES))
@pytest.mark.parametrize("kind", list(gu.KINDS))
def test_golden_reference_outputs(wa, gpu, tmp_path, name, kind):
    idx, data = gu.build_index(wa, name, kind, tmp_path)
    K = int(data["meta"][3])
    Q = data["Q"]
    nq = Q.shape[0]
    n = 0
    for key, method, beam, mult, p in gu.cases(data, kind):
        if not SUPPORTED(kind, method):
            continue
        args = (Q, data["W_" + p], nq) + ((method,) if kind.endswith("RangeFilterTreeIndex") else ())
        ids, dists = idx.batch_search(*args, _qp(wa, beam, mult, K))
        tie = kind in gu.TIE_AWARE_KINDS or p in ("-7", "edge") or method in ("fenwick", "three_split")
        ctx = gu.RowContext(data["X"], data["labels"], Q, data["W_" + p], gu.metric_of(gu.FIXTURES[name]))
        ok, why = gu.same_rows(data["ids|" + key], data["dists|" + key], ids, dists, tie, ctx)
        assert ok, f"{name} {key}: {why}"
        n += 1
    assert n > 0


def test_golden_ratio_fallback(wa, gpu, tmp_path):
    """min_query_to_bucket_ratio (src/range_filter_tree.h:460-466) against the REAL reference's rows (ratio_golden.npz)"""
    n, failures = gu.replay_ratio(wa, tmp_path)
    assert n == 48
    assert not failures, "\n".join(failures[:10])


@pytest.mark.parametrize("name", list(gu.FIXTURES))
def test_golden_quirks(wa, gpu, tmp_path, name):
    idx, data = gu.build_index(wa, name, "VamanaRangeFilterTreeIndex", tmp_path)
    Q, W = data["Q"], data["W_-3"]
    nq = Q.shape[0]
    ids, dists = idx.batch_search(Q, W, nq, "optimized_postfilter", _qp(wa, 64, 1, 10, max_beam=64))
    assert np.array_equal(ids, data["ids|VamanaRangeFilterTreeIndex|maxbeam"])
    assert np.array_equal(dists, data["dists|VamanaRangeFilterTreeIndex|maxbeam"])
    ids, dists = idx.batch_search(Q, W, nq, "optimized_postfilter", _qp(wa, 8, 1, 10, max_beam=20))
    assert np.array_equal(ids, data["ids|VamanaRangeFilterTreeIndex|overshoot"])
    assert np.array_equal(dists, data["dists|VamanaRangeFilterTreeIndex|overshoot"])


# ------------------------------------------------------------------------------------------
# product-built indexes against the oracle on larger seeded inputs (graphs shared via the cache)
# ------------------------------------------------------------------------------------------
CASES = [
    ("VamanaRangeFilterTreeIndex", "FloatEuclidian", sift_like, 128, 6000, dict(cutoff=500, split_factor=2)),
    ("VamanaRangeFilterTreeIndex", "FloatMips", unit_mixture, 100, 5000, dict(cutoff=400, split_factor=3)),
    ("VamanaRangeFilterTreeIndex", "FloatMips", unit_mixture, 96, 6000, dict(cutoff=300, split_factor=4)),  # deep-like 4-WST
    ("SuperOptimizedPostfilterTreeIndex", "FloatMips", unit_mixture, 100, 5000, dict(cutoff=400, split_factor=2, shift_factor=0.5)),
    ("SuperOptimizedPostfilterTreeIndex", "FloatEuclidian", sift_like, 96, 5000, dict(cutoff=300, split_factor=2.5, shift_factor=0.3)),
    ("PostfilterVamanaIndex", "FloatEuclidian", sift_like, 64, 4000, dict()),
    ("RangeFilterTreeIndex", "FloatEuclidian", sift_like, 48, 4000, dict(cutoff=300, split_factor=2)),
    ("PrefilterIndex", "FloatMips", unit_mixture, 100, 4000, dict()),
    # byte rows (python_bindings.cpp:234-237): dimensions beyond what float32 accumulation represents exactly
    ("VamanaRangeFilterTreeIndex", "UInt8Euclidian", sift_like, 300, 4000, dict(cutoff=400, split_factor=2)),
    ("SuperOptimizedPostfilterTreeIndex", "Int8Mips", lambda n, d, s: (lambda m, g=sift_like(n, d, s): g(m) - 128.0), 520, 3000,
     dict(cutoff=300, split_factor=2, shift_factor=0.5)),
    ("VamanaRangeFilterTreeIndex", "Int8Euclidian", lambda n, d, s: (lambda m, g=sift_like(n, d, s): g(m) - 128.0), 70, 3000, dict(cutoff=300, split_factor=3)),
    ("PrefilterIndex", "UInt8Mips", sift_like, 200, 3000, dict()),
]


@pytest.mark.parametrize("kind,sfx,gen,d,n,kw", CASES)
def test_index_matches_oracle(oracle, wa, gpu, tmp_path, kind, sfx, gen, d, n, kw):
    nq = 300
    g = gen(n, d, 21)
    X, Q = g(n), g(nq)
    labels = distinct_labels(n, 3)
    cache = str(tmp_path) + "/"
    labkw = "filters" if kind == "PostfilterVamanaIndex" else "filter_values"
    pi = getattr(wa, kind + sfx)(X, **{labkw: labels}, build_params=wa.BuildParams(32, 64, 1.0, cache), **kw)
    oi = getattr(oracle, kind + sfx)(X, **{labkw: labels}, build_params=oracle.BuildParams(32, 64, 1.0, cache), **kw)
    tree = kind.endswith("RangeFilterTreeIndex")
    fractions = [-9, -6, -4, -2, -1, 0] if kind != "PrefilterIndex" else [-6, -3, -1]
    for p in fractions:
        W = windows(labels, nq, p, seed=50 + p)
        for beam, mult in [(10, 1), (20, 3), (80, 1), (200, 2)]:
            a = (Q, W, nq) + (("optimized_postfilter",) if tree else ())
            ids, dists = pi.batch_search(*a, _qp(wa, beam, mult))
            eids, edists = oi.batch_search(*a, _qp(oracle, beam, mult))
            tie = kind in gu.TIE_AWARE_KINDS or p <= -6
            ok, why = gu.same_rows(eids, edists, ids, dists, tie, gu.RowContext(X, labels, Q, W, gu.metric_of(sfx)))
            assert ok, f"{kind}{sfx} p={p} beam={beam} mult={mult}: {why}"
            c, oc = pi.counters(), oi.last_counters
            assert c["beam_searches"] == oc["searches"] and c["hops"] == oc["hops"]
            assert c["dist_cmps"] + c["brute_rows"] == oc["dist_cmps"]


def test_edge_cases(oracle, wa, gpu, tmp_path):
    n, d, nq = 3000, 32, 64
    g = sift_like(n, d, 9)
    X, Q = g(n), g(nq)
    labels = distinct_labels(n, 4)
    cache = str(tmp_path) + "/"
    pi = wa.VamanaRangeFilterTreeIndexFloatEuclidian(X, labels, cutoff=400, split_factor=2, build_params=wa.BuildParams(24, 48, 1.0, cache))
    oi = oracle.VamanaRangeFilterTreeIndexFloatEuclidian(X, labels, cutoff=400, split_factor=2, build_params=oracle.BuildParams(24, 48, 1.0, cache))
    s = np.sort(labels)
    W = np.zeros((nq, 2))
    W[0::4] = (5.0, 6.0)                 # outside the label span -> padding
    W[1::4] = (s[10], s[10])             # zero width -> empty index range
    W[2::4] = (s[100], s[103])           # fewer than k points -> padded tail
    W[3::4] = (s[0] - 1, s[-1] + 1)      # everything
    for beam, mult, mb in [(10, 1, 10000), (16, 2, 10000), (16, 1, 40), (50, 4, 120)]:
        ids, dists = pi.batch_search(Q, W, nq, "optimized_postfilter", _qp(wa, beam, mult, 10, mb))
        eids, edists = oi.batch_search(Q, W, nq, "optimized_postfilter", _qp(oracle, beam, mult, 10, mb))
        ok, why = gu.same_rows(eids, edists, ids, dists, True, gu.RowContext(X, labels, Q, W, "l2"))
        assert ok, why
    assert (ids[0] == 0).all() and (dists[0] == FLT_MAX).all()
    # empty batch and k = 1 / k = 37
    ids, dists = pi.batch_search(Q[:0], W[:0], 0, "optimized_postfilter", _qp(wa, 10))
    assert ids.shape == (0, 10)
    for k in (1, 37):
        ids, dists = pi.batch_search(Q, W, nq, "optimized_postfilter", _qp(wa, 40, 1, k))
        eids, edists = oi.batch_search(Q, W, nq, "optimized_postfilter", _qp(oracle, 40, 1, k))
        ok, why = gu.same_rows(eids, edists, ids, dists, True, gu.RowContext(X, labels, Q, W, "l2"))
        assert ok, why
    # errors: bad shapes raise RuntimeError like the reference (tree_utils.h:46-60)
    with pytest.raises(RuntimeError):
        wa.VamanaRangeFilterTreeIndexFloatEuclidian(X.reshape(-1), labels)
    with pytest.raises(RuntimeError):
        wa.VamanaRangeFilterTreeIndexFloatEuclidian(X, labels[:-1])


def test_device_resident_call_matches_host_call(wa, gpu, tmp_path):
    torch = pytest.importorskip("torch")
    n, d, nq = 4000, 64, 500
    g = sift_like(n, d, 2)
    X, Q = g(n), g(nq)
    labels = distinct_labels(n, 5)
    idx = wa.VamanaRangeFilterTreeIndexFloatEuclidian(X, labels, cutoff=500, split_factor=2, build_params=wa.BuildParams(32, 64, 1.0, ""))
    W = windows(labels, nq, -3, 1).astype(np.float32)
    qp = _qp(wa, 20, 2)
    ids, dists = idx.batch_search(Q, W, nq, "optimized_postfilter", qp)
    dev = torch.device("cuda:0")
    tq, tw = torch.from_numpy(Q).to(dev), torch.from_numpy(W).to(dev)
    tids = torch.empty((nq, 10), dtype=torch.int32, device=dev)
    tdist = torch.empty((nq, 10), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    idx.batch_search_device(tq.data_ptr(), tw.data_ptr(), nq, 0, "optimized_postfilter", qp, tids.data_ptr(), tdist.data_ptr(), 0)
    assert np.array_equal(tids.cpu().numpy().view(np.uint32), ids)
    assert np.array_equal(tdist.cpu().numpy(), dists)
    # a shard keeps its global query numbering (query row number = "own id" quirk)
    half = nq // 2
    idx.batch_search_device(tq[half:].data_ptr(), tw[half:].data_ptr(), nq - half, half, "optimized_postfilter", qp,
                            tids[half:].data_ptr(), tdist[half:].data_ptr(), 0)
    assert np.array_equal(tids.cpu().numpy().view(np.uint32), ids)


def test_asynchronous_calls_return_the_blocking_calls_rows(wa, gpu):
    """wann_batch_search_device_async / wann_wait (ABI 4): batches in flight two at a time on the index's lanes -- different
    windows, beams and (mid-fraction) scheduling machinery in neighbouring batches -- must give the rows and the work counters of
    the blocking call, whatever overlaps with whatever; a PrefilterIndex (whose dense-path buffers belong to the index) too."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda:0")
    n, d, nq = 30000, 64, 1500
    g = sift_like(n, d, 12)
    X, Q = g(n), g(nq)
    labels = distinct_labels(n, 13)
    idx = wa.VamanaRangeFilterTreeIndexFloatEuclidian(X, labels, cutoff=500, split_factor=2, build_params=wa.BuildParams(32, 64, 1.0, ""))
    pre = wa.PrefilterIndexFloatEuclidian(X, labels)
    tq = torch.from_numpy(Q).to(dev)
    batches = [(p, beam, mult) for p in (-7, -3, -5, -1, -9, -4, -6, -2) for beam, mult in ((20, 2),)] + [(-8, 10, 1), (-3, 40, 1)]
    for index, method in ((idx, "optimized_postfilter"), (pre, "")):
        want, outs = [], []
        for i, (p, beam, mult) in enumerate(batches):
            W = windows(labels, nq, p, 100 + i).astype(np.float32)
            tw = torch.from_numpy(W).to(dev)
            ti = torch.empty((nq, 10), dtype=torch.int32, device=dev)
            td = torch.empty((nq, 10), dtype=torch.float32, device=dev)
            torch.cuda.synchronize()
            index.batch_search_device(tq.data_ptr(), tw.data_ptr(), nq, 0, method, _qp(wa, beam, mult), ti.data_ptr(), td.data_ptr(), 0)
            c = index.counters()
            want.append((ti.cpu().numpy().copy(), td.cpu().numpy().copy(), {k_: c[k_] for k_ in ("beam_searches", "hops", "dist_cmps", "brute_rows")}))
            outs.append((tw, torch.zeros_like(ti), torch.zeros_like(td)))
        torch.cuda.synchronize()
        for rep in range(3):
            tickets = []
            for i, (p, beam, mult) in enumerate(batches):
                tw, ti, td = outs[i]
                ti.zero_()
                td.zero_()
                tickets.append(index.batch_search_device_async(tq.data_ptr(), tw.data_ptr(), nq, 0, method, _qp(wa, beam, mult), ti.data_ptr(), td.data_ptr(), 0))
                if i >= 1:  # two in flight: wait for ticket i - 1 before submitting i + 1
                    c = index.wait(tickets[i - 1])
                    wi, wd, wc = want[i - 1]
                    assert np.array_equal(outs[i - 1][1].cpu().numpy(), wi), (rep, i - 1, batches[i - 1])
                    assert np.array_equal(outs[i - 1][2].cpu().numpy(), wd), (rep, i - 1, batches[i - 1])
                    assert {k_: c[k_] for k_ in wc} == wc, (rep, i - 1, batches[i - 1])
            c = index.wait(tickets[-1])
            assert np.array_equal(outs[-1][1].cpu().numpy(), want[-1][0]) and np.array_equal(outs[-1][2].cpu().numpy(), want[-1][1])
    with pytest.raises(RuntimeError):
        idx.wait(10 ** 6)  # no such ticket


def test_default_stream_call_is_ordered_with_queued_torch_work(wa, gpu):
    """Stream 0 = the HIP default stream: the call is ordered after torch work still queued there (exact
    ground truth by GEMM + topk, whose freed temporaries the caching allocator hands out again as the
    engine's output rows) and the host-buffer call must return the same rows.  This interleaving used to
    end in GPU memory faults (k_route needed a scratch segment; NULL meant a private non-blocking stream)."""
    torch = pytest.importorskip("torch")
    n, d, nq = 60000, 100, 4000
    g = unit_mixture(n, d, 2025)
    X, Q = g(n), g(nq)
    labels = distinct_labels(n, 77)
    idx = wa.SuperOptimizedPostfilterTreeIndexFloatMips(X, labels, cutoff=1000, split_factor=2, shift_factor=0.5,
                                                        build_params=wa.BuildParams(32, 100, 1.0, ""))
    dev = torch.device("cuda:0")
    Xt, labt, Qt = torch.from_numpy(X).to(dev), torch.from_numpy(labels).to(dev), torch.from_numpy(Q).to(dev)
    qp = _qp(wa, 10, 2)
    for it in range(6):
        W = windows(labels, nq, -6, it).astype(np.float32)
        Wt = torch.from_numpy(W).to(dev)
        gt = torch.empty((nq, 10), dtype=torch.int64, device=dev)
        for a in range(0, nq, 256):  # queued, not waited for
            s = -(Qt[a:a + 256] @ Xt.T)
            s.masked_fill_(~((labt[None, :] >= Wt[a:a + 256, 0:1]) & (labt[None, :] <= Wt[a:a + 256, 1:2])), float("inf"))
            gt[a:a + 256] = torch.topk(s, 10, dim=1, largest=False).indices
        del s
        ids_t = torch.empty((nq, 10), dtype=torch.int32, device=dev)
        dist_t = torch.empty((nq, 10), dtype=torch.float32, device=dev)
        idx.batch_search_device(Qt.data_ptr(), Wt.data_ptr(), nq, 0, "", qp, ids_t.data_ptr(), dist_t.data_ptr(), 0)
        got = ids_t.to(torch.int64) & 0xFFFFFFFF
        rec = float((gt[:, :, None] == got[:, None, :]).any(2).sum(1).double().mean().item() / 10)
        assert rec > 0.9, (it, rec)
        ids, dists = idx.batch_search(Q, W, nq, qp)
        assert np.array_equal(got.cpu().numpy().astype(np.uint32), ids)
        assert np.array_equal(dist_t.cpu().numpy(), dists)
        del ids_t, dist_t, gt, Wt


# ------------------------------------------------------------------------------------------
# GPU Vamana build: byte-identical graph files to the host builder (hence to the oracle's builder)
# ------------------------------------------------------------------------------------------
BUILD_CASES = [
    ("VamanaRangeFilterTreeIndex", "FloatEuclidian", sift_like, 128, 6000, dict(cutoff=500, split_factor=2), 32, 64),
    ("VamanaRangeFilterTreeIndex", "FloatMips", unit_mixture, 100, 4000, dict(cutoff=400, split_factor=2), 24, 48),
    ("SuperOptimizedPostfilterTreeIndex", "FloatEuclidian", sift_like, 40, 3000, dict(cutoff=300, split_factor=2, shift_factor=0.5), 16, 100),
    ("PostfilterVamanaIndex", "FloatEuclidian", sift_like, 64, 20000, dict(), 64, 500),
    ("VamanaRangeFilterTreeIndex", "UInt8Euclidian", sift_like, 512, 3000, dict(cutoff=400, split_factor=2), 24, 48),  # byte rows
]


@pytest.mark.parametrize("kind,sfx,gen,d,n,kw,R,L", BUILD_CASES)
def test_gpu_builder_matches_host_builder(wa, gpu, tmp_path, monkeypatch, kind, sfx, gen, d, n, kw, R, L):
    import os
    X = gen(n, d, 77)(n)
    labels = distinct_labels(n, 13)
    labkw = "filters" if kind == "PostfilterVamanaIndex" else "filter_values"
    gdir, hdir = str(tmp_path / "gpu") + "/", str(tmp_path / "host") + "/"
    os.makedirs(gdir), os.makedirs(hdir)
    monkeypatch.delenv("WANN_HOST_BUILD", raising=False)
    getattr(wa, kind + sfx)(X, **{labkw: labels}, build_params=wa.BuildParams(R, L, 1.0, gdir), **kw)
    monkeypatch.setenv("WANN_HOST_BUILD", "1")
    getattr(wa, kind + sfx)(X, **{labkw: labels}, build_params=wa.BuildParams(R, L, 1.0, hdir), **kw)
    gf, hf = sorted(os.listdir(gdir)), sorted(os.listdir(hdir))
    assert gf == hf and len(gf) > 0
    for f in gf:
        a, b = open(gdir + f, "rb").read(), open(hdir + f, "rb").read()
        assert a == b, f"{f}: GPU-built graph differs from the host-built one"


def test_gpu_builder_restarts_with_a_larger_visited_buffer(wa, gpu, tmp_path, monkeypatch, capfd):
    """A visited list that outgrows its LDS buffer restarts the build on the GPU (no host fallback): with the
    buffer forced down to 64 entries the first attempts overflow and the final graph is still the golden one."""
    import os
    data = gu.load_build()
    name = "gauss_l2"
    X, labels, (R, L, metric) = data[f"{name}|X"], data[f"{name}|labels"], data[f"{name}|meta"]
    cdir = str(tmp_path) + "/"
    monkeypatch.delenv("WANN_HOST_BUILD", raising=False)
    monkeypatch.setenv("WANN_BUILD_VIS_CAP", "64")
    monkeypatch.setenv("WANN_VERBOSE", "1")
    wa.PostfilterVamanaIndexFloatEuclidian(X, labels, wa.BuildParams(int(R), int(L), 1.0, cdir))
    assert "restarting the GPU build" in capfd.readouterr().err
    assert open(cdir + os.listdir(cdir)[0], "rb").read() == data[f"{name}|file"].tobytes()


@pytest.mark.parametrize("name", gu.BUILD_CASES)
def test_gpu_builder_writes_the_references_graph_file(wa, gpu, tmp_path, monkeypatch, name):
    """The graph the GPU builder writes equals, name and bytes, the file the REAL reference's builder wrote
    for the same input (continuous coordinates: no exactly equidistant candidates)."""
    import os
    data = gu.load_build()
    X, labels, (R, L, metric) = data[f"{name}|X"], data[f"{name}|labels"], data[f"{name}|meta"]
    cdir = str(tmp_path) + "/"
    monkeypatch.delenv("WANN_HOST_BUILD", raising=False)
    cls = wa.PostfilterVamanaIndexFloatMips if metric else wa.PostfilterVamanaIndexFloatEuclidian
    cls(X, labels, wa.BuildParams(int(R), int(L), 1.0, cdir))
    assert os.listdir(cdir) == [data[f"{name}|file_name"].tobytes().decode()]
    assert open(cdir + os.listdir(cdir)[0], "rb").read() == data[f"{name}|file"].tobytes()


def test_gpu_builder_reference_tie_order(wa, gpu, tmp_path, monkeypatch):
    """Integer-valued vectors: prunes and neighbour sorts are full of exactly equidistant candidates, which the reference
    orders with std::sort on the distance alone.  With WANN_REF_TIES=1 the GPU builder runs the restated std::sort
    (wann_stdsort.h) on the same sequences and (1) writes the file the REAL reference wrote for the integer-valued golden
    input, (2) agrees with the host builder (which calls std::sort itself) on a SIFT-like tree of graphs -- uint8 rows too."""
    import os
    data = gu.load_build()
    name = "int_l2"
    X, labels, (R, L, metric) = data[f"{name}|X"], data[f"{name}|labels"], data[f"{name}|meta"]
    monkeypatch.setenv("WANN_REF_TIES", "1")
    monkeypatch.delenv("WANN_HOST_BUILD", raising=False)
    cdir = str(tmp_path / "golden") + "/"
    os.makedirs(cdir)
    wa.PostfilterVamanaIndexFloatEuclidian(X, labels, wa.BuildParams(int(R), int(L), 1.0, cdir))
    assert os.listdir(cdir) == [data[f"{name}|file_name"].tobytes().decode()]
    assert open(cdir + os.listdir(cdir)[0], "rb").read() == data[f"{name}|file"].tobytes()
    monkeypatch.setenv("WANN_REF_TIES", "0")
    ddir = str(tmp_path / "default") + "/"
    os.makedirs(ddir)
    wa.PostfilterVamanaIndexFloatEuclidian(X, labels, wa.BuildParams(int(R), int(L), 1.0, ddir))
    assert open(ddir + os.listdir(ddir)[0], "rb").read() != data[f"{name}|file"].tobytes()  # (ties by id: another valid graph)
    monkeypatch.setenv("WANN_REF_TIES", "1")
    for sfx, d, n in (("FloatEuclidian", 128, 5000), ("UInt8Euclidian", 64, 3000)):
        X2 = sift_like(n, d, 78)(n)
        lab2 = distinct_labels(n, 14)
        gdir, hdir = str(tmp_path / ("gpu" + sfx)) + "/", str(tmp_path / ("host" + sfx)) + "/"
        os.makedirs(gdir), os.makedirs(hdir)
        monkeypatch.delenv("WANN_HOST_BUILD", raising=False)
        getattr(wa, "VamanaRangeFilterTreeIndex" + sfx)(X2, lab2, cutoff=500, split_factor=2, build_params=wa.BuildParams(24, 48, 1.0, gdir))
        monkeypatch.setenv("WANN_HOST_BUILD", "1")
        getattr(wa, "VamanaRangeFilterTreeIndex" + sfx)(X2, lab2, cutoff=500, split_factor=2, build_params=wa.BuildParams(24, 48, 1.0, hdir))
        gf, hf = sorted(os.listdir(gdir)), sorted(os.listdir(hdir))
        assert gf == hf and len(gf) > 0
        for f in gf:
            assert open(gdir + f, "rb").read() == open(hdir + f, "rb").read(), f"{sfx} {f}: GPU and host builders differ in reference tie order"


# ------------------------------------------------------------------------------------------
# multi-bucket query methods (fenwick, three_split) and the ratio fallback, against the oracle
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("sfx,gen,d,split,cutoff", [("FloatEuclidian", sift_like, 64, 2, 300), ("FloatMips", unit_mixture, 100, 3, 250),
                                                    ("FloatEuclidian", sift_like, 32, 6, 400)])
def test_fenwick_and_three_split_match_oracle(oracle, wa, gpu, tmp_path, sfx, gen, d, split, cutoff):
    n, nq = 5000, 200
    g = gen(n, d, 41)
    X, Q = g(n), g(nq)
    labels = distinct_labels(n, 8)
    cache = str(tmp_path) + "/"
    kw = dict(cutoff=cutoff, split_factor=split)
    for kind in ("VamanaRangeFilterTreeIndex", "RangeFilterTreeIndex"):
        pi = getattr(wa, kind + sfx)(X, labels, build_params=wa.BuildParams(24, 48, 1.0, cache), **kw)
        oi = getattr(oracle, kind + sfx)(X, labels, build_params=oracle.BuildParams(24, 48, 1.0, cache), **kw)
        for p in (-9, -6, -4, -2, -1, 0):
            W = windows(labels, nq, p, seed=70 + p)
            for method in ("fenwick", "three_split", "smart_combined", "optimized_postfilter"):
                for beam, mult, ratio in [(10, 1, None), (20, 3, None), (20, 2, 1.5)]:
                    ids, dists = pi.batch_search(Q, W, nq, method, _qp(wa, beam, mult, ratio=ratio))
                    eids, edists = oi.batch_search(Q, W, nq, method, _qp(oracle, beam, mult, ratio=ratio))
                    ok, why = gu.same_rows(eids, edists, ids, dists, True, gu.RowContext(X, labels, Q, W, gu.metric_of(sfx)))
                    assert ok, f"{kind}{sfx} split={split} p={p} {method} beam={beam} x{mult} ratio={ratio}: {why}"
                    if kind.startswith("Vamana"):
                        c, oc = pi.counters(), oi.last_counters
                        assert c["beam_searches"] == oc["searches"] and c["hops"] == oc["hops"], (p, method, beam, mult, ratio)


def test_scheduling_variants_return_identical_rows(wa, gpu, tmp_path, monkeypatch):
    """Speculative concurrent doubling, the register-resident beam and the clash-check fast paths are
    scheduling / layout choices: every combination must return the same rows and the same counters."""
    n, d, nq = 20000, 128, 600
    g = sift_like(n, d, 4)
    X, Q = g(n), g(nq)
    labels = distinct_labels(n, 6)
    idx = wa.VamanaRangeFilterTreeIndexFloatEuclidian(X, labels, cutoff=1000, split_factor=2, build_params=wa.BuildParams(32, 100, 1.0, ""))
    base = {}
    for env in ({}, {"WANN_NO_SPEC": "1"}, {"WANN_FORCE_GENERAL": "1"}, {"WANN_NO_SPEC": "1", "WANN_FORCE_GENERAL": "1"}, {"WANN_OLD_GENERAL": "1"}):
        for k_ in ("WANN_NO_SPEC", "WANN_FORCE_GENERAL", "WANN_OLD_GENERAL"):
            monkeypatch.delenv(k_, raising=False)
        for k_, v in env.items():
            monkeypatch.setenv(k_, v)
        for p in (-7, -5, -3, 0):
            W = windows(labels, nq, p, seed=90 + p)
            for beam, mult in [(10, 1), (40, 2), (80, 1), (100, 1)]:
                ids, dists = idx.batch_search(Q, W, nq, "optimized_postfilter", _qp(wa, beam, mult))
                c = idx.counters()
                key = (p, beam, mult)
                cur = (ids.copy(), dists.copy(), c["beam_searches"], c["hops"], c["dist_cmps"])
                if key not in base:
                    base[key] = cur
                else:
                    b = base[key]
                    assert np.array_equal(b[0], cur[0]) and np.array_equal(b[1], cur[1]), (env, key)
                    assert b[2:] == cur[2:], (env, key, b[2:], cur[2:])


def test_big_workgroup_levels_return_identical_rows(wa, gpu, monkeypatch):
    """Windows that are a tiny share of their partition double far beyond the in-kernel cap (1280): those levels
    are searched speculatively by single waves that own a whole workgroup's LDS.  With that path off
    (WANN_NO_BIG: sequential follow-up launches) or all speculation off the rows and the counters must not change."""
    n, d, nq = 150000, 64, 1500
    g = sift_like(n, d, 14)
    X, Q = g(n), g(nq)
    labels = distinct_labels(n, 16)
    idx = wa.VamanaRangeFilterTreeIndexFloatEuclidian(X, labels, cutoff=1000, split_factor=2, build_params=wa.BuildParams(24, 64, 1.0, ""))
    W = windows(labels, nq, -9, seed=5)
    base = {}
    for env in ({}, {"WANN_NO_BIG": "1"}, {"WANN_NO_SPEC": "1"}, {"WANN_OLD_GENERAL": "1"}):
        for k_ in ("WANN_NO_SPEC", "WANN_NO_BIG", "WANN_OLD_GENERAL"):
            monkeypatch.delenv(k_, raising=False)
        for k_, v in env.items():
            monkeypatch.setenv(k_, v)
        for beam, mult in [(10, 1), (80, 2)]:
            ids, dists = idx.batch_search(Q, W, nq, "optimized_postfilter", _qp(wa, beam, mult))
            c = idx.counters()
            cur = (ids.copy(), dists.copy(), c["beam_searches"], c["hops"], c["dist_cmps"])
            if not env:
                assert c["beam_searches"] > 1.5 * nq  # the doubling loop is really exercised
                base[(beam, mult)] = cur
            else:
                b = base[(beam, mult)]
                assert np.array_equal(b[0], cur[0]) and np.array_equal(b[1], cur[1]), (env, beam, mult)
                assert b[2:] == cur[2:], (env, beam, mult, b[2:], cur[2:])
                if "WANN_NO_BIG" in env and beam == 10:
                    assert c["rounds"] >= 2  # some task doubled beyond the cap: the follow-up launch ran


def _continuation_case(wa):
    n, d, nq = 150000, 64, 1500
    g = sift_like(n, d, 14)
    X, Q = g(n), g(nq)
    labels = distinct_labels(n, 16)
    idx = wa.VamanaRangeFilterTreeIndexFloatEuclidian(X, labels, cutoff=1000, split_factor=2, build_params=wa.BuildParams(24, 64, 1.0, ""))
    return idx, Q, labels, nq


@pytest.mark.parametrize("variant", ["default", "WANN_LA_EAGER", "WANN_NO_LOOKAHEAD"])
def test_mid_fraction_machinery_matches_oracle(oracle, wa, gpu, tmp_path, monkeypatch, variant):
    """The scheduling machinery of the mid window fractions -- speculative levels, the companion launch of the one-wave kernel
    (search wave + scoring helper waves), pollers, look-aheads, deep hand-offs -- against the ORACLE (not against itself) at
    a size where it runs: windows that are 1/256 ... 1/512 of the root partition double up to beam 5 120.  Rows and the work
    counters (searches, hops, dist_cmps) must equal the oracle's; the batch-level counters prove that the paths ran."""
    n, d, nq = 150000, 64, 800
    g = sift_like(n, d, 14)
    X, Q = g(n), g(nq)
    labels = distinct_labels(n, 16)
    cache = str(tmp_path) + "/"
    if variant == "WANN_NO_LOOKAHEAD":  # (no look-ahead searches and no scan of the idle pollers)
        monkeypatch.setenv("WANN_NO_LOOKAHEAD", "1")
    if variant == "WANN_LA_EAGER":  # (every chain that fails its second level asks for a look-ahead)
        monkeypatch.setenv("WANN_LA_EAGER", "1")
    monkeypatch.setenv("WANN_DEEP_MIN_TASKS", "1")  # (800 tasks would not count as a saturated launch)
    idx = wa.VamanaRangeFilterTreeIndexFloatEuclidian(X, labels, cutoff=1000, split_factor=2, build_params=wa.BuildParams(24, 64, 1.0, cache))
    ref = oracle.VamanaRangeFilterTreeIndexFloatEuclidian(X, labels, cutoff=1000, split_factor=2, build_params=oracle.BuildParams(24, 64, 1.0, cache))
    tot = dict(rounds=0, spec_searches=0, big_searches=0, packet_hops=0, deep_handoffs=0, lookaheads_used=0, lookaheads_issued=0)
    for p, beam, mult in [(-9, 10, 1), (-9, 40, 2), (-8, 80, 1), (-8, 10, 2), (-6, 40, 1), (-6, 80, 2), (-3, 64, 1), (-4, 64, 2)]:
        W = windows(labels, nq, p, seed=5)
        ids, dists = idx.batch_search(Q, W, nq, "optimized_postfilter", _qp(wa, beam, mult))
        c = idx.counters()
        eids, edists = ref.batch_search(Q, W, nq, "optimized_postfilter", _qp(oracle, beam, mult))
        oc = ref.last_counters
        assert np.array_equal(dists, edists), (p, beam, mult, np.nonzero((dists != edists).any(1))[0][:10])
        assert np.array_equal(ids, eids), (p, beam, mult, np.nonzero((ids != eids).any(1))[0][:10])
        assert (c["beam_searches"], c["hops"], c["dist_cmps"]) == (oc["searches"], oc["hops"], oc["dist_cmps"]), (p, beam, mult, c, oc)
        assert c["recovered_continuations"] == 0 and c["poll_timeouts"] == 0
        for k_ in tot:
            tot[k_] += c[k_]
    assert tot["spec_searches"] > 0, tot       # speculative levels were searched
    assert tot["big_searches"] > 0, tot        # the companion launch's one-wave kernel ran searches ...
    assert tot["packet_hops"] > 0, tot         # ... fed by its helper waves
    if variant == "WANN_LA_EAGER":
        assert tot["lookaheads_issued"] > 0 and tot["lookaheads_used"] > 0, tot  # pollers searched levels ahead, and chains took them
    print(variant, tot)


def test_final_research_of_a_resolved_chain_is_never_moot(oracle, wa, gpu, tmp_path):
    """final_beam_multiply > 1: the wave that resolves a doubling chain goes on AS THE PARENT and searches the final beam.  Its
    sub-task's moot state (a lower level of the same parent succeeded meanwhile) must not follow it there -- it once did, and a
    third of the runs of this batch lost a row's final search (timing dependent, hence the repetitions).  Rows and work counters
    against the oracle, every time."""
    n, d, nq = 150000, 64, 1500
    g = sift_like(n, d, 14)
    X, Q = g(n), g(nq)
    labels = distinct_labels(n, 16)
    cache = str(tmp_path) + "/"
    idx = wa.VamanaRangeFilterTreeIndexFloatEuclidian(X, labels, cutoff=1000, split_factor=2, build_params=wa.BuildParams(24, 64, 1.0, cache))
    ref = oracle.VamanaRangeFilterTreeIndexFloatEuclidian(X, labels, cutoff=1000, split_factor=2, build_params=oracle.BuildParams(24, 64, 1.0, cache))
    for p, beam, mult, reps in [(-8, 20, 3, 12), (-9, 40, 2, 6)]:
        W = windows(labels, nq, p, seed=5)
        eids, edists = ref.batch_search(Q, W, nq, "optimized_postfilter", _qp(oracle, beam, mult))
        oc = ref.last_counters
        for rep in range(reps):
            ids, dists = idx.batch_search(Q, W, nq, "optimized_postfilter", _qp(wa, beam, mult))
            c = idx.counters()
            assert np.array_equal(dists, edists) and np.array_equal(ids, eids), (p, beam, mult, rep, np.nonzero((ids != eids).any(1))[0][:10])
            assert (c["beam_searches"], c["hops"], c["dist_cmps"]) == (oc["searches"], oc["hops"], oc["dist_cmps"]), (p, beam, mult, rep, c, oc)
        assert c["spec_searches"] > 0


def test_deep_chains_go_to_idle_pollers(wa, gpu, monkeypatch):
    """A saturated launch ends with its few longest doubling chains.  With enough tasks a handful of pollers (a CU each) take
    over chains that reach their third level; which wave runs a search must not change a row or a work counter."""
    idx, Q, labels, nq = _continuation_case(wa)
    handed = stranded = 0
    # (2^-5 / 2^-6 at beam 64: windows too narrow for the extra speculated level of short chains -- k_route, round 6 -- so their
    # third levels are sequential continuations, which is what gets handed over)
    for p, beam, mult in [(-3, 64, 1), (-5, 64, 1), (-4, 64, 2), (-6, 64, 2), (-2, 100, 1)]:
        narrow = p <= -5  # (such a batch may also have chains that double beyond the in-kernel cap: companion mode, whose unserved
        #                    continuations the host recovers -- test_unserved_continuations_are_recovered)
        W = windows(labels, nq, p, seed=9)
        monkeypatch.setenv("WANN_DEEP_MIN_TASKS", str(10**9))  # (no launch of this test counts as saturated: no deep-chain pollers)
        ids, dists = idx.batch_search(Q, W, nq, "optimized_postfilter", _qp(wa, beam, mult))
        c = idx.counters()
        assert c["deep_handoffs"] == 0
        monkeypatch.delenv("WANN_DEEP_MIN_TASKS")
        monkeypatch.setenv("WANN_DEEP_MIN_TASKS", "1")
        ids2, dists2 = idx.batch_search(Q, W, nq, "optimized_postfilter", _qp(wa, beam, mult))
        c2 = idx.counters()
        monkeypatch.delenv("WANN_DEEP_MIN_TASKS")
        assert np.array_equal(ids, ids2) and np.array_equal(dists, dists2), (p, beam, mult)
        assert (c["beam_searches"], c["hops"], c["dist_cmps"]) == (c2["beam_searches"], c2["hops"], c2["dist_cmps"]), (c, c2)
        assert c2["recovered_continuations"] == 0
        handed += c2["deep_handoffs"]
        # pollers that give up at once (a profiler serialising the launches) strand what was handed to them: the host
        # re-queues it, same rows, same counters
        monkeypatch.setenv("WANN_DEEP_MIN_TASKS", "1")
        monkeypatch.setenv("WANN_FORCE_POLL_TIMEOUT", "1")
        ids3, dists3 = idx.batch_search(Q, W, nq, "optimized_postfilter", _qp(wa, beam, mult))
        c3 = idx.counters()
        monkeypatch.delenv("WANN_DEEP_MIN_TASKS")
        monkeypatch.delenv("WANN_FORCE_POLL_TIMEOUT")
        assert np.array_equal(ids, ids3) and np.array_equal(dists, dists3), (p, beam, mult)
        assert (c["beam_searches"], c["hops"], c["dist_cmps"]) == (c3["beam_searches"], c3["hops"], c3["dist_cmps"]), (c, c3)
        # nothing is handed to pollers that are not there: no hand-off, no look-ahead, nothing for the host to recover
        assert c3["deep_handoffs"] == 0 and c3["lookaheads_issued"] == 0 and (narrow or c3["recovered_continuations"] == 0), c3
        stranded += c3["recovered_continuations"]
    assert handed > 0, "no chain of this test reached its third level next to an idle poller"


def test_lookahead_searches_change_nothing(wa, gpu, monkeypatch):
    """Chains that keep failing have levels searched ahead by idle pollers (and move to pollers themselves).  Batch after batch on
    one index -- the per-task look-ahead slots are state that must not leak from one batch into the next -- rows and work
    counters equal those of the plain sequential chains."""
    idx, Q, labels, nq = _continuation_case(wa)
    used = 0
    for rep in range(2):
        for p, beam, mult in [(-9, 10, 1), (-6, 40, 1), (-8, 20, 2), (-6, 40, 1), (-3, 64, 1)]:
            W = windows(labels, nq, p, seed=5 + rep)
            monkeypatch.setenv("WANN_DEEP_MIN_TASKS", "1")
            if rep == 1:
                monkeypatch.setenv("WANN_LA_EAGER", "1")
            ids, dists = idx.batch_search(Q, W, nq, "optimized_postfilter", _qp(wa, beam, mult))
            c = idx.counters()
            monkeypatch.delenv("WANN_LA_EAGER", raising=False)
            monkeypatch.setenv("WANN_NO_LOOKAHEAD", "1")
            ids2, dists2 = idx.batch_search(Q, W, nq, "optimized_postfilter", _qp(wa, beam, mult))
            c2 = idx.counters()
            monkeypatch.delenv("WANN_NO_LOOKAHEAD")
            monkeypatch.delenv("WANN_DEEP_MIN_TASKS")
            assert c2["lookaheads_used"] == 0
            assert np.array_equal(ids, ids2) and np.array_equal(dists, dists2), (rep, p, beam, mult)
            assert (c["beam_searches"], c["hops"], c["dist_cmps"]) == (c2["beam_searches"], c2["hops"], c2["dist_cmps"]), (c, c2)
            assert c["recovered_continuations"] == 0
            used += c["lookaheads_used"]
    assert used > 0, "no look-ahead was taken in this test"


def test_unserved_continuations_are_recovered(wa, gpu, monkeypatch):
    """Tasks that must double beyond the in-kernel cap after their speculative levels failed are handed to pollers of the
    companion launch.  Pollers that give up (launches serialised by the runtime or a profiler) leave them in the hand-over
    list; the host re-queues them for the follow-up launch: same rows, same counters, no error."""
    idx, Q, labels, nq = _continuation_case(wa)
    seen_recovery = 0
    for p, beam, mult in [(-9, 10, 1), (-8, 20, 1), (-7, 40, 2)]:
        W = windows(labels, nq, p, seed=5)
        monkeypatch.delenv("WANN_FORCE_POLL_TIMEOUT", raising=False)
        ids, dists = idx.batch_search(Q, W, nq, "optimized_postfilter", _qp(wa, beam, mult))
        c = idx.counters()
        assert c["recovered_continuations"] == 0
        monkeypatch.setenv("WANN_FORCE_POLL_TIMEOUT", "1")
        ids2, dists2 = idx.batch_search(Q, W, nq, "optimized_postfilter", _qp(wa, beam, mult))
        c2 = idx.counters()
        monkeypatch.delenv("WANN_FORCE_POLL_TIMEOUT")
        assert np.array_equal(ids, ids2) and np.array_equal(dists, dists2), (p, beam, mult)
        assert (c["beam_searches"], c["hops"], c["dist_cmps"]) == (c2["beam_searches"], c2["hops"], c2["dist_cmps"])
        seen_recovery += c2["recovered_continuations"]
    assert seen_recovery > 0, "no batch of this test produced a continuation: the recovery path did not run"


def test_serialised_kernel_dispatch_is_survived(wa, gpu, tmp_path):
    """AMD_SERIALIZE_KERNEL=3 makes the runtime run one kernel at a time: the companion launch's pollers can then never
    meet their producers.  Default behaviour: pollers are not used at all under that setting; with WANN_FORCE_POLLERS the
    pollers run, notice that the other launch has not started, give up, and the host recovers.  Rows must equal an
    ordinary run's in both cases (child processes: the runtime reads the variable at start-up)."""
    import subprocess
    import sys
    script = tmp_path / "serial.py"
    script.write_text(
        "import os, sys, numpy as np\n"
        f"sys.path.insert(0, {REPO!r}); sys.path.insert(0, os.path.join({REPO!r}, 'tests'))\n"
        "import rangefilteredann_amd, window_ann as wa\n"
        "from util import sift_like, distinct_labels, windows\n"
        "n, d, nq = 150000, 64, 1500\n"
        "g = sift_like(n, d, 14); X, Q = g(n), g(nq); labels = distinct_labels(n, 16)\n"
        "idx = wa.VamanaRangeFilterTreeIndexFloatEuclidian(X, labels, cutoff=1000, split_factor=2, build_params=wa.BuildParams(24, 64, 1.0, sys.argv[1]))\n"
        "W = windows(labels, nq, -9, seed=5)\n"
        "ids, dists = idx.batch_search(Q, W, nq, 'optimized_postfilter', wa.QueryParams(10, 10, 1.35, 10**7, 10**4, 1, 10000, None, False))\n"
        "c = idx.counters()\n"
        "np.savez(sys.argv[2], ids=ids, dists=dists, rec=c['recovered_continuations'], searches=c['beam_searches'])\n")
    cache = str(tmp_path / "cache") + "/"
    os.makedirs(cache)
    outs = {}
    for name, env in (("plain", {}), ("serial", {"AMD_SERIALIZE_KERNEL": "3"}), ("serial_pollers", {"AMD_SERIALIZE_KERNEL": "3", "WANN_FORCE_POLLERS": "1"})):
        e = dict(os.environ)
        e.update(env)
        r = subprocess.run([sys.executable, str(script), cache, str(tmp_path / (name + ".npz"))], env=e, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, (name, r.stderr[-2000:])
        outs[name] = np.load(tmp_path / (name + ".npz"))
    for name in ("serial", "serial_pollers"):
        assert np.array_equal(outs[name]["ids"], outs["plain"]["ids"]) and np.array_equal(outs[name]["dists"], outs["plain"]["dists"]), name
        assert int(outs[name]["searches"]) == int(outs["plain"]["searches"])
    assert int(outs["serial"]["rec"]) == 0  # pollers off: nothing to recover


def test_in_process_multi_device_mode(wa, gpu, tmp_path, monkeypatch):
    """WANN_DEVICES: the index is replicated per listed device and the host-buffer batch_search -- the call the reference's driver
    makes (run_our_method.py unchanged) -- cuts its batch into contiguous shards with global query numbers.  On the one-GPU
    box the list names device 0 twice (then three times, with an odd batch): rows and work counters equal the single call's."""
    n, d, nq = 20000, 64, 777
    g = sift_like(n, d, 21)
    X, Q = g(n), g(nq)
    labels = distinct_labels(n, 22)
    cache = str(tmp_path) + "/"
    mk = lambda: wa.VamanaRangeFilterTreeIndexFloatEuclidian(X, labels, cutoff=500, split_factor=2, build_params=wa.BuildParams(24, 48, 1.0, cache))  # noqa: E731
    single = mk()
    assert single.num_replicas() == 1
    for devs in ("0,0", "0,0,0"):
        monkeypatch.setenv("WANN_DEVICES", devs)
        multi = mk()
        monkeypatch.delenv("WANN_DEVICES")
        assert multi.num_replicas() == len(devs.split(","))
        for p, method, beam, mult in [(-3, "optimized_postfilter", 20, 2), (-8, "optimized_postfilter", 10, 1), (-5, "fenwick", 10, 1), (-12, "three_split", 10, 1)]:
            W = windows(labels, nq, p, seed=33)
            # (queries 0..nq-1 carry ids that name nodes of the root partition: the own-id skip must see GLOBAL numbers in every shard)
            i1, d1 = single.batch_search(Q, W, nq, method, _qp(wa, beam, mult))
            c1 = single.counters()
            i2, d2 = multi.batch_search(Q, W, nq, method, _qp(wa, beam, mult))
            c2 = multi.counters()
            assert np.array_equal(d1, d2), (devs, p, method)
            if method == "optimized_postfilter" and p > -12:
                assert np.array_equal(i1, i2), (devs, p, method)
            assert (c1["beam_searches"], c1["hops"], c1["dist_cmps"], c1["brute_rows"]) == (c2["beam_searches"], c2["hops"], c2["dist_cmps"], c2["brute_rows"]), (devs, p, method)
        del multi


def test_c_abi_end_to_end_through_ctypes(oracle, wa, gpu, tmp_path):
    """include/wann.h bound with ctypes exactly as INTEGRATION.md shows for a foreign host: create, search (host
    buffers), counters, destroy -- against the oracle."""
    import ctypes as C
    import rangefilteredann_amd
    lib = C.CDLL(rangefilteredann_amd.lib_path())

    class QP(C.Structure):
        _fields_ = [("k", C.c_int64), ("beam_width", C.c_int64), ("cut", C.c_double), ("limit", C.c_int64), ("degree_limit", C.c_int64),
                    ("final_beam_multiply", C.c_int64), ("postfiltering_max_beam", C.c_int64), ("has_ratio", C.c_int32), ("ratio", C.c_float),
                    ("verbose", C.c_int32)]

    class BP(C.Structure):
        _fields_ = [("max_degree", C.c_int64), ("limit", C.c_int64), ("alpha", C.c_double), ("cache_path", C.c_char_p)]

    class CTR(C.Structure):
        _fields_ = [(f, C.c_int64) for f in ("beam_searches", "hops", "dist_cmps", "brute_rows", "label_reads", "rounds", "spec_searches",
                                            "spec_hops", "spec_dist_cmps", "gemm_queries")] + [("device_ms", C.c_double), ("search_kernel_ms", C.c_double),
                                                                                              ("recovered_continuations", C.c_int64), ("gemm_unproven", C.c_int64), ("gemm_rescued", C.c_int64), ("deep_handoffs", C.c_int64), ("lookaheads_used", C.c_int64),
                    ("big_searches", C.c_int64), ("big_hops", C.c_int64), ("packet_hops", C.c_int64), ("own_scorings", C.c_int64), ("prefetched_hops", C.c_int64), ("poll_timeouts", C.c_int64), ("lookaheads_issued", C.c_int64)]

    lib.wann_index_create.restype = C.c_void_p
    lib.wann_index_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int32, C.c_double, C.c_double,
                                      C.POINTER(BP), C.c_int, C.c_int]
    lib.wann_batch_search.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_char_p, C.POINTER(QP), C.c_void_p, C.c_void_p]
    lib.wann_get_counters.argtypes = [C.c_void_p, C.POINTER(CTR)]
    lib.wann_index_destroy.argtypes = [C.c_void_p]
    lib.wann_last_error.restype = C.c_char_p
    n, d, nq, k = 5000, 48, 200, 10
    g = sift_like(n, d, 31)
    X, Q = np.ascontiguousarray(g(n)), np.ascontiguousarray(g(nq))
    labels = distinct_labels(n, 7)
    cache = (str(tmp_path) + "/").encode()
    bp = BP(24, 48, 1.0, cache)
    h = lib.wann_index_create(3, 0, 0, X.ctypes.data, n, d, labels.ctypes.data, 400, 2.0, 0.5, C.byref(bp), 0, 0)
    assert h, lib.wann_last_error()
    oi = oracle.VamanaRangeFilterTreeIndexFloatEuclidian(X, labels, cutoff=400, split_factor=2, build_params=oracle.BuildParams(24, 48, 1.0, cache.decode()))
    try:
        for p, method in ((-3, "optimized_postfilter"), (-5, "fenwick"), (-2, "three_split")):
            W = windows(labels, nq, p, seed=3)
            W32 = np.ascontiguousarray(W.astype(np.float32))
            ids = np.zeros((nq, k), dtype=np.uint32)
            dists = np.zeros((nq, k), dtype=np.float32)
            qp = QP(k, 20, 1.35, 10**7, 10**4, 2, 10000, 0, 0.0, 0)
            rc = lib.wann_batch_search(h, Q.ctypes.data, W32.ctypes.data, nq, method.encode(), C.byref(qp), ids.ctypes.data, dists.ctypes.data)
            assert rc == 0, lib.wann_last_error()
            eids, edists = oi.batch_search(Q, W, nq, method, _qp(oracle, 20, 2))
            ok, why = gu.same_rows(eids, edists, ids, dists, method != "optimized_postfilter", gu.RowContext(X, labels, Q, W, "l2"))
            assert ok, (method, why)
            ctr = CTR()
            assert lib.wann_get_counters(h, C.byref(ctr)) == 0
            assert ctr.beam_searches == oi.last_counters["searches"] and ctr.hops == oi.last_counters["hops"]
        bad = QP(0, 20, 1.35, 10**7, 10**4, 2, 10000, 0, 0.0, 0)
        assert lib.wann_batch_search(h, Q.ctypes.data, W32.ctypes.data, nq, b"fenwick", C.byref(bad), ids.ctypes.data, dists.ctypes.data) != 0
        assert b"k must be" in lib.wann_last_error()
    finally:
        lib.wann_index_destroy(h)


def test_predicted_costs_follow_the_work(wa, gpu):
    """wann_predict_costs (what a cost-balanced shard cut balances): deterministic, positive, and its batch total tracks the
    work the search then does (hops + scanned rows / 36) across window fractions within a small factor; a cut of equal
    predicted work is a valid cut"""
    from rangefilteredann_amd.distributed import weighted_bounds
    n, d, nq = 40000, 32, 600
    g = sift_like(n, d, 51)
    X, Q = g(n), g(nq)
    labels = distinct_labels(n, 17)
    idx = wa.VamanaRangeFilterTreeIndexFloatEuclidian(X, labels, cutoff=500, split_factor=2, build_params=wa.BuildParams(32, 64, 1.0, ""))
    qp = _qp(wa, 20, 1)
    tot = {}
    for p in (-1, -4, -6, -8, -11):
        W = windows(labels, nq, p, 60 + p).astype(np.float32)
        c1 = idx.predict_costs(W, "optimized_postfilter", qp)
        c2 = idx.predict_costs(W, "optimized_postfilter", qp)
        assert c1.shape == (nq,) and np.array_equal(c1, c2) and (c1 > 0).all()
        idx.batch_search(Q, W, nq, "optimized_postfilter", qp)
        c = idx.counters()
        work = c["hops"] + c["spec_hops"] * 0 + c["brute_rows"] / 36.0 + nq * (c["brute_rows"] > 0)
        tot[p] = (float(c1.sum()), work)
        assert 0.33 < tot[p][0] / max(work, 1.0) < 3.0, (p, tot[p])
        b = weighted_bounds(c1, 8)
        assert b[0][0] == 0 and b[-1][1] == nq and all(b[i][1] == b[i + 1][0] for i in range(7))
        assert max(float(c1[lo:hi].sum()) for lo, hi in b) <= float(c1.sum()) / 8 + float(c1.max()) + 1e-3
    f = idx.predict_costs(windows(labels, nq, -5, 1).astype(np.float32), "fenwick", qp)  # (multi-task queries)
    assert (f > 0).all()


def test_c_abi_allgather_leaves_device_resident_rows(wa, gpu, tmp_path):
    """wann_batch_search_allgather (ABI 4): the shard search writes into the all-gather's send planes, ONE ncclAllGather (RCCL,
    opened with dlopen) leaves [replicas][2][cap][k] planes on the device, and the rows read back from them equal
    wann_batch_search's.  One GPU here: a group of one rank -- the communicator, the collective and the plane layout are real."""
    import ctypes as C
    import rangefilteredann_amd
    lib = C.CDLL(rangefilteredann_amd.lib_path())
    hip = C.CDLL("libamdhip64.so")

    class QP(C.Structure):
        _fields_ = [("k", C.c_int64), ("beam_width", C.c_int64), ("cut", C.c_double), ("limit", C.c_int64), ("degree_limit", C.c_int64),
                    ("final_beam_multiply", C.c_int64), ("postfiltering_max_beam", C.c_int64), ("has_ratio", C.c_int32), ("ratio", C.c_float),
                    ("verbose", C.c_int32)]

    class BP(C.Structure):
        _fields_ = [("max_degree", C.c_int64), ("limit", C.c_int64), ("alpha", C.c_double), ("cache_path", C.c_char_p)]

    lib.wann_index_create.restype = C.c_void_p
    lib.wann_index_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int32, C.c_double, C.c_double,
                                      C.POINTER(BP), C.c_int, C.c_int]
    lib.wann_batch_search.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_char_p, C.POINTER(QP), C.c_void_p, C.c_void_p]
    lib.wann_batch_search_allgather.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_char_p, C.POINTER(QP), C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
    lib.wann_gather_layout.argtypes = [C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.wann_num_replicas.argtypes = [C.c_void_p]
    lib.wann_index_destroy.argtypes = [C.c_void_p]
    lib.wann_last_error.restype = C.c_char_p
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    n, d, nq, k = 6000, 32, 333, 10
    g = sift_like(n, d, 41)
    X, Q = np.ascontiguousarray(g(n)), np.ascontiguousarray(g(nq))
    labels = distinct_labels(n, 9)
    bp = BP(24, 48, 1.0, (str(tmp_path) + "/").encode())
    h = lib.wann_index_create(3, 0, 0, X.ctypes.data, n, d, labels.ctypes.data, 400, 2.0, 0.5, C.byref(bp), 0, 0)
    assert h, lib.wann_last_error()
    try:
        G = lib.wann_num_replicas(h)
        for p in (-2, -6):
            W32 = np.ascontiguousarray(windows(labels, nq, p, seed=5).astype(np.float32))
            qp = QP(k, 20, 1.35, 10**7, 10**4, 2, 10000, 0, 0.0, 0)
            ids = np.zeros((nq, k), dtype=np.uint32)
            dists = np.zeros((nq, k), dtype=np.float32)
            assert lib.wann_batch_search(h, Q.ctypes.data, W32.ctypes.data, nq, b"optimized_postfilter", C.byref(qp), ids.ctypes.data, dists.ctypes.data) == 0
            planes = (C.c_void_p * G)()
            cap = C.c_int64(0)
            rc = lib.wann_batch_search_allgather(h, Q.ctypes.data, W32.ctypes.data, nq, b"optimized_postfilter", C.byref(qp), planes, C.byref(cap))
            assert rc == 0, lib.wann_last_error()
            assert cap.value == (nq + G - 1) // G
            for r in range(G):  # every replica's device holds every shard's planes
                host = np.zeros((G, 2, cap.value, k), dtype=np.int32)
                assert hip.hipMemcpy(host.ctypes.data, planes[r], host.nbytes, 2) == 0
                for s_ in range(G):
                    lo, cnt = C.c_int64(0), C.c_int64(0)
                    assert lib.wann_gather_layout(nq, G, s_, C.byref(lo), C.byref(cnt), None) == 0
                    assert np.array_equal(host[s_, 0, :cnt.value].view(np.uint32), ids[lo.value:lo.value + cnt.value])
                    assert np.array_equal(host[s_, 1, :cnt.value].view(np.float32), dists[lo.value:lo.value + cnt.value])
                    assert (host[s_, 0, cnt.value:] == 0).all() and (host[s_, 1, cnt.value:].view(np.float32) == np.finfo(np.float32).max).all()
    finally:
        lib.wann_index_destroy(h)


# ------------------------------------------------------------------------------------------
# dense prefilter path (queries sharing a window -> MFMA GEMM + exact re-rank), adversarial-style data
# (generate_datasets/generate_advserial_dataset.py:8-69: clusters, labels c - 0.5 + U(0,1), one window per cluster)
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("metric,sfx,d", [("mips", "FloatMips", 100), ("l2", "FloatEuclidian", 128), ("l2f", "FloatEuclidian", 96),
                                          ("mips", "FloatMips", 512), ("mips", "FloatMips", 300), ("l2f", "FloatEuclidian", 200),
                                          ("l2", "FloatEuclidian", 384)])  # (129 .. 512: the wide kernel, 2 / 3 / 4 slabs, partial last slab)
def test_dense_prefilter_matches_oracle(oracle, wa, gpu, monkeypatch, metric, sfx, d):
    rng = np.random.default_rng(17)
    nclu, per, qper = 12, 1500, 40
    n = nclu * per
    if metric == "l2":
        X = sift_like(n, d, 3)(n)
        Q = sift_like(n, d, 3)(nclu * qper)
    else:
        cent = rng.standard_normal((nclu, d))
        X = (cent[np.repeat(np.arange(nclu), per)] + 0.3 * rng.standard_normal((n, d)))
        Q = (cent[np.repeat(np.arange(nclu), qper)] + 0.3 * rng.standard_normal((nclu * qper, d)))
        X = (X / np.linalg.norm(X, axis=1, keepdims=True)).astype(np.float32)
        Q = (Q / np.linalg.norm(Q, axis=1, keepdims=True)).astype(np.float32)
    labels = (np.repeat(np.arange(nclu), per) - 0.5 + rng.random(n)).astype(np.float32)
    perm = rng.permutation(n)
    X, labels = X[perm], labels[perm]
    nq = Q.shape[0]
    W = np.zeros((nq, 2))
    cl = np.repeat(np.arange(nclu), qper)
    W[:, 0], W[:, 1] = cl - 0.5, cl + 0.5
    W[::7] = (2.2, 3.9)       # a second family of shared windows
    W[5::31, 1] += 1e-3 * np.arange(len(W[5::31]))  # and some unique ones (exact scan)
    pi = getattr(wa, "PrefilterIndex" + sfx)(X, labels)
    oi = getattr(oracle, "PrefilterIndex" + sfx)(X, labels)
    for k in (10, 16):
        monkeypatch.delenv("WANN_NO_GEMM", raising=False)
        ids, dists = pi.batch_search(Q, W, nq, _qp(wa, 10, 1, k))
        c = pi.counters()
        assert c["gemm_queries"] > nq // 2, c
        eids, edists = oi.batch_search(Q, W, nq, _qp(oracle, 10, 1, k))
        ok, why = gu.same_rows(eids, edists, ids, dists, True, gu.RowContext(X, labels, Q, W, gu.metric_of(sfx)))
        assert ok, f"{metric} k={k}: {why}"
        monkeypatch.setenv("WANN_NO_GEMM", "1")
        ids2, dists2 = pi.batch_search(Q, W, nq, _qp(wa, 10, 1, k))
        assert pi.counters()["gemm_queries"] == 0
        assert np.array_equal(ids, ids2) and np.array_equal(dists, dists2)


def test_dense_prefilter_is_skipped_on_streams_without_shared_windows(wa, gpu, monkeypatch):
    """After two batches in a row without any window group the dense path is only tried every eighth batch (its launches are a
    tenth of a tiny-window batch); when shared windows come back it picks up again.  Rows never depend on which path ran."""
    rng = np.random.default_rng(23)
    n, d, nq = 30000, 32, 400
    X = rng.standard_normal((n, d)).astype(np.float32)
    Q = rng.standard_normal((nq, d)).astype(np.float32)
    labels = rng.permutation(n).astype(np.float32)
    pi = wa.PrefilterIndexFloatEuclidian(X, labels)
    distinct = np.stack([np.arange(nq) * 10.0 + 0.5, np.arange(nq) * 10.0 + 3000.5], 1)  # 3 000 points each, all different
    shared = np.tile(np.array([[100.5, 9100.5]]), (nq, 1))
    monkeypatch.setenv("WANN_NO_GEMM", "1")
    want_d = pi.batch_search(Q, distinct, nq, _qp(wa, 10, 1, 10))
    want_s = pi.batch_search(Q, shared, nq, _qp(wa, 10, 1, 10))
    monkeypatch.delenv("WANN_NO_GEMM")
    for _ in range(3):
        got = pi.batch_search(Q, distinct, nq, _qp(wa, 10, 1, 10))
        assert pi.counters()["gemm_queries"] == 0
        assert np.array_equal(got[0], want_d[0]) and np.array_equal(got[1], want_d[1])
    dense = []
    for _ in range(10):
        got = pi.batch_search(Q, shared, nq, _qp(wa, 10, 1, 10))
        dense.append(pi.counters()["gemm_queries"])
        assert np.array_equal(got[0], want_s[0]) and np.array_equal(got[1], want_s[1])
    assert dense[0] == 0 and max(dense) == nq, dense          # skipped at first, picked up within eight batches ...
    assert dense[dense.index(nq):] == [nq] * (10 - dense.index(nq)), dense  # ... and then on every batch
    monkeypatch.setenv("WANN_DENSE_ALWAYS", "1")
    pi2 = wa.PrefilterIndexFloatEuclidian(X, labels)
    for _ in range(3):
        pi2.batch_search(Q, distinct, nq, _qp(wa, 10, 1, 10))
    pi2.batch_search(Q, shared, nq, _qp(wa, 10, 1, 10))
    assert pi2.counters()["gemm_queries"] == nq


@pytest.mark.parametrize("sfx,d,style", [("FloatMips", 100, "unit"), ("FloatEuclidian", 128, "sift"), ("FloatEuclidian", 24, "unit"),
                                         ("FloatMips", 7, "drift"), ("FloatMips", 512, "unit"), ("FloatEuclidian", 260, "unit")])
def test_dense_prefilter_slices_and_tiles(wa, gpu, monkeypatch, sfx, d, style):
    """Windows longer than one slice (2 048 positions), longer than eight (the slice grows), groups of more than 128 queries
    (several tiles), duplicates (equal scores), and labels that correlate with the geometry (the best candidates all sit at
    the end of the window: every list overflows again and again).  The MFMA path must return exactly what the exact scan does."""
    rng = np.random.default_rng(5)
    n = 42000
    if style == "sift":
        X = sift_like(n, d, 3)(n)
        Q = sift_like(n, d, 3)(700)
    else:
        X = rng.standard_normal((n, d)).astype(np.float32)
        Q = rng.standard_normal((700, d)).astype(np.float32)
        if style == "unit":
            X /= np.linalg.norm(X, axis=1, keepdims=True)
    X[1000:1040] = X[1000]  # duplicates
    labels = rng.permutation(n).astype(np.float32)
    if style == "drift":  # sorted by the score of the first query family: later labels = better candidates
        order = np.argsort(X @ Q[0])
        labels = np.empty(n, dtype=np.float32)
        labels[order] = np.arange(n, dtype=np.float32)
        Q[:] = Q[0] * (1 + 0.01 * rng.standard_normal((700, 1))).astype(np.float32)
    nq = Q.shape[0]
    W = np.zeros((nq, 2))
    W[:300] = (100.5, 25100.5)       # 25 000 positions: eight slices of 3 200; three query tiles
    W[300:520] = (30000.5, 35000.5)  # 5 000 positions: three slices; two tiles
    W[520:560] = (-1, 1e9)           # everything
    W[560:660] = (500.5, 1800.5)     # 1 300 positions: one short slice
    W[660:] = (7.5, 250.5)           # below the minimum window: exact scan
    pi = getattr(wa, "PrefilterIndex" + sfx)(X, labels)
    for k in (10, 1):
        monkeypatch.delenv("WANN_NO_GEMM", raising=False)
        ids, dists = pi.batch_search(Q, W, nq, _qp(wa, 10, 1, k))
        c = pi.counters()
        assert c["gemm_queries"] == 660, c
        monkeypatch.setenv("WANN_NO_GEMM", "1")
        ids2, dists2 = pi.batch_search(Q, W, nq, _qp(wa, 10, 1, k))
        assert pi.counters()["gemm_queries"] == 0
        assert np.array_equal(dists, dists2), (style, k, int((dists != dists2).any(axis=1).sum()), c)
        assert np.array_equal(ids, ids2), (style, k, int((ids != ids2).any(axis=1).sum()), c)
        if style == "unit" and d == 100:
            assert c["gemm_unproven"] < 66, c  # the proof must hold for nearly every query on well-separated data
        if style == "drift" and k == 10:
            # the best points sit next to each other in label order (the same 64-position blocks): the exact scan of
            # those few blocks settles such queries, not the scan of the whole window
            assert c["gemm_rescued"] > 300 and c["gemm_unproven"] < 100, c


@pytest.mark.parametrize("sfx,k", [("FloatEuclidian", 10), ("FloatMips", 100), ("UInt8Euclidian", 7)])
def test_short_scan_lists_are_split(oracle, wa, gpu, monkeypatch, sfx, k):
    """A few queries over big windows: the exact scan cuts every window into slices (one wave each, the last one merges).
    Results must equal the oracle's and the unsplit scan's, also for k larger than a slice's share and for empty slices."""
    rng = np.random.default_rng(23)
    n, d, nq = 30000, 20, 5
    if sfx.startswith("UInt8"):
        X = rng.integers(0, 256, (n, d)).astype(np.uint8)
        Q = rng.integers(0, 256, (nq, d)).astype(np.uint8)
    else:
        X = rng.standard_normal((n, d)).astype(np.float32)
        Q = rng.standard_normal((nq, d)).astype(np.float32)
    labels = rng.permutation(n).astype(np.float32)
    W = np.array([[-1, 1e9], [10.5, 29000.5], [5.5, 105.5], [7.5, 9.5], [100.5, 20100.5]])
    pi = getattr(wa, "PrefilterIndex" + sfx)(X, labels)
    oi = getattr(oracle, "PrefilterIndex" + sfx)(X, labels)
    monkeypatch.delenv("WANN_NO_SPLIT_SCAN", raising=False)
    ids, dists = pi.batch_search(Q, W, nq, _qp(wa, 10, 1, k))
    eids, edists = oi.batch_search(Q, W, nq, _qp(oracle, 10, 1, k))
    ok, why = gu.same_rows(eids, edists, ids, dists, True, gu.RowContext(X, labels, Q, W, gu.metric_of(sfx)))
    assert ok, why
    monkeypatch.setenv("WANN_NO_SPLIT_SCAN", "1")
    ids2, dists2 = pi.batch_search(Q, W, nq, _qp(wa, 10, 1, k))
    assert np.array_equal(ids, ids2) and np.array_equal(dists, dists2)


@pytest.mark.parametrize("n,d", [(1, 4), (2, 3), (7, 5), (300, 5), (300, 17)])
def test_tiny_shapes(oracle, wa, gpu, n, d):
    """Degenerate sizes: single-point partitions, dimensions that are not multiples of 4 / 8, windows wider
    than the data."""
    rng = np.random.default_rng(n * 31 + d)
    X = rng.integers(0, 50, size=(n, d)).astype(np.float32)
    Q = rng.integers(0, 50, size=(9, d)).astype(np.float32)
    labels = distinct_labels(n, 1)
    W = np.array([[-1.0, 2.0], [0.0, 0.5], [0.5, 1.0], [0.2, 0.21], [labels[0], labels[0]], [3.0, 4.0], [-2.0, -1.0], [0.0, 1.0], [0.4, 0.6]])
    for kind, kw in (("VamanaRangeFilterTreeIndex", dict(cutoff=50, split_factor=2)), ("SuperOptimizedPostfilterTreeIndex", dict(cutoff=50, split_factor=2, shift_factor=0.5)),
                     ("PostfilterVamanaIndex", {}), ("RangeFilterTreeIndex", dict(cutoff=50, split_factor=2))):
        if n < 7 and kind.endswith("RangeFilterTreeIndex"):
            continue  # the reference's fenwick lookup indexes past its bucket table on such trees (std::out_of_range)
        for sfx in ("FloatEuclidian", "FloatMips"):
            labkw = "filters" if kind == "PostfilterVamanaIndex" else "filter_values"
            pi = getattr(wa, kind + sfx)(X, **{labkw: labels}, build_params=wa.BuildParams(8, 16, 1.0, ""), **kw)
            oi = getattr(oracle, kind + sfx)(X, **{labkw: labels}, build_params=oracle.BuildParams(8, 16, 1.0, ""), **kw)
            for beam, mult in ((1, 1), (4, 3), (16, 1)):
                a = (Q, W, 9) + (("optimized_postfilter",) if kind.endswith("RangeFilterTreeIndex") else ())
                ids, dists = pi.batch_search(*a, _qp(wa, beam, mult, 3))
                eids, edists = oi.batch_search(*a, _qp(oracle, beam, mult, 3))
                ok, why = gu.same_rows(eids, edists, ids, dists, True, gu.RowContext(X, labels, Q, W, gu.metric_of(sfx)))
                assert ok, f"n={n} d={d} {kind}{sfx} beam={beam} x{mult}: {why}"


def test_prefilter_classes_accept_beam_zero(oracle, wa, gpu):
    """The reference driver calls the brute-force classes with beam_size = 0 (run_our_method.py:256)."""
    n, d, nq = 1500, 24, 20
    g = sift_like(n, d, 5)
    X, Q = g(n), g(nq)
    labels = distinct_labels(n, 2)
    W = windows(labels, nq, -3, 1)
    for cls, args in (("PrefilterIndexFloatEuclidian", ()), ("RangeFilterTreeIndexFloatEuclidian", ("fenwick",))):
        ids, dists = getattr(wa, cls)(X, labels).batch_search(Q, W, nq, *args, _qp(wa, 0, 1, 10))
        eids, edists = getattr(oracle, cls)(X, labels).batch_search(Q, W, nq, *args, _qp(oracle, 0, 1, 10))
        ok, why = gu.same_rows(eids, edists, ids, dists, True, gu.RowContext(X, labels, Q, W, "l2"))
        assert ok, f"{cls}: {why}"
    with pytest.raises(RuntimeError, match="beam_width must be positive"):
        wa.PostfilterVamanaIndexFloatEuclidian(X, labels, wa.BuildParams(8, 16, 1.0, "")).batch_search(Q, W, nq, _qp(wa, 0, 1, 10))


def test_per_query_ids_and_levels_dealt_search(wa, gpu, oracle, tmp_path):
    """wann_batch_search_device_ids (ABI 5): a non-contiguous subset of a batch, every query under its own global row number, returns
    that subset's rows of the whole batch's call (the reference's own-id quirk, beamSearch.h:128: rows depend on the number); and
    level_dealt_batch_search over it -- single doubling levels as items of their own, the sequential rule afterwards -- returns the
    plain call's rows for any prediction of the level counts (postfilter_vamana.h:161-181)."""
    import torch
    from rangefilteredann_amd.distributed import level_dealt_batch_search
    n, d, nq, k = 6000, 32, 120, 10
    g = sift_like(n, d, 41)
    X, Q = g(n), g(nq)
    labels = distinct_labels(n, 42)
    W = np.concatenate([windows(labels, nq, p, 70 + p)[i::5] for i, p in enumerate((-8, -6, -4, -3, -1))])[:nq].astype(np.float32)
    idx = wa.VamanaRangeFilterTreeIndexFloatEuclidian(X, labels, cutoff=400, split_factor=2, build_params=wa.BuildParams(32, 64, 1.0, str(tmp_path) + "/"))
    dev = torch.device("cuda:0")
    Qt, Wt = torch.from_numpy(Q).to(dev), torch.from_numpy(W).to(dev)
    # ids of the queries: some name nodes of the partitions they search (row numbers ARE small integers: the quirk is live)
    for beam, mult, max_beam in ((5, 1, 10000), (10, 3, 10000), (5, 2, 70)):
        qp = wa.QueryParams(k, beam, 1.35, 10**7, 10**4, mult, max_beam, None, False)
        eids, edists = idx.batch_search(Q, W, nq, "optimized_postfilter", qp)
        sel = torch.tensor(sorted(np.random.default_rng(beam).choice(nq, 50, replace=False).tolist()), dtype=torch.int64, device=dev)
        oi = torch.empty((len(sel), k), dtype=torch.int32, device=dev)
        od = torch.empty((len(sel), k), dtype=torch.float32, device=dev)
        qsel, wsel = Qt[sel].contiguous(), Wt[sel].contiguous()  # (kept alive: a temporary's block goes back to the allocator at once)
        idx.batch_search_device_ids(qsel.data_ptr(), wsel.data_ptr(), len(sel), sel.data_ptr(), "optimized_postfilter", qp, oi.data_ptr(), od.data_ptr(), 0)
        torch.cuda.synchronize()
        assert np.array_equal(oi.cpu().numpy().view(np.uint32), eids[sel.cpu().numpy()]) and np.array_equal(od.cpu().numpy(), edists[sel.cpu().numpy()])

        def run_group(qn, b, mb, m):
            qs, ws = Qt[qn].contiguous(), Wt[qn].contiguous()
            ri = torch.empty((len(qn), k), dtype=torch.int32, device=dev)
            rd = torch.empty((len(qn), k), dtype=torch.float32, device=dev)
            idx.batch_search_device_ids(qs.data_ptr(), ws.data_ptr(), len(qn), qn.contiguous().data_ptr(), "optimized_postfilter",
                                        wa.QueryParams(k, b, 1.35, 10**7, 10**4, m, mb, None, False), ri.data_ptr(), rd.data_ptr(), 0)
            return ri, rd
        levels = np.random.default_rng(100 + beam).integers(1, 7, nq).tolist()
        ids, dists = level_dealt_batch_search(run_group, nq, k, beam, max_beam, mult, levels, device=dev)
        torch.cuda.synchronize()
        assert np.array_equal(dists.cpu().numpy(), edists), (beam, mult, max_beam)
        assert np.array_equal(ids.cpu().numpy().view(np.uint32), eids), (beam, mult, max_beam)
