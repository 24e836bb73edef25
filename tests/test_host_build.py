"""CPU: the product's host index builder (C ABI wann_build_cache_shard, no GPU needed) against the
oracle's builder: graph cache files must be byte-identical, the on-disk format is the reference's
(graph.h:174-196) and splitting the build over shards yields the same files."""
import os

import numpy as np
import pytest

from util import distinct_labels, sift_like, unit_mixture

CASES = [
    ("tree_l2", 3, 0, sift_like, 64, 3000, dict(cutoff=300, split_factor=2)),
    ("tree_mips_b3", 3, 1, unit_mixture, 100, 2500, dict(cutoff=300, split_factor=3)),
    ("super_mips", 4, 1, unit_mixture, 48, 2500, dict(cutoff=300, split_factor=2, shift_factor=0.5)),
    ("postfilter_l2", 1, 0, sift_like, 32, 2000, dict()),
]
ORC_CLS = {3: "VamanaRangeFilterTreeIndex", 4: "SuperOptimizedPostfilterTreeIndex", 1: "PostfilterVamanaIndex"}


def _read(d):
    return {f: open(os.path.join(d, f), "rb").read() for f in sorted(os.listdir(d))}


@pytest.mark.parametrize("name,kind,metric,gen,d,n,kw", CASES)
def test_builder_matches_oracle_builder(oracle, wa, tmp_path, name, kind, metric, gen, d, n, kw):
    X = gen(n, d, 17)(n)
    labels = distinct_labels(n, 6)
    pdir, odir = str(tmp_path / "product") + "/", str(tmp_path / "oracle") + "/"
    os.makedirs(pdir), os.makedirs(odir)
    R, L = 24, 48
    wa.build_cache_shard(kind, metric, X, labels, kw.get("cutoff", 1000), kw.get("split_factor", 2),
                         kw.get("shift_factor", 0.5), wa.BuildParams(R, L, 1.0, pdir), 0, 1, 4)
    sfx = "FloatMips" if metric else "FloatEuclidian"
    labkw = "filters" if kind == 1 else "filter_values"
    oidx = getattr(oracle, ORC_CLS[kind] + sfx)(X, **{labkw: labels}, build_params=oracle.BuildParams(R, L, 1.0, odir), threads=4, **kw)
    pf, of = _read(pdir), _read(odir)
    assert list(pf) == list(of) and len(pf) == sum(oidx.levels())
    for f in pf:
        assert pf[f] == of[f], f"graph file {f} differs"
    # on-disk format: [n][maxDeg][deg...][edges...] int32, readable by the oracle's loader
    f0 = sorted(pf)[0]
    rows = oracle.graph_load(os.path.join(pdir, f0))
    hdr = np.frombuffer(pf[f0][:8], dtype=np.int32)
    assert hdr[0] == rows.shape[0] and hdr[1] == R == rows.shape[1] - 1
    assert (rows[:, 0] <= R).all() and (rows[:, 0] >= 0).all()


@pytest.mark.parametrize("name", ["gauss_l2", "unit_mips"])
def test_builder_writes_the_references_graph_file(wa, tmp_path, name):
    """Continuous inputs (no exactly equidistant candidates): the cache file equals the one the REAL
    reference's builder wrote (tests/golden/build_golden.npz), name and bytes."""
    import golden_util as gu
    data = gu.load_build()
    X, labels, (R, L, metric) = data[f"{name}|X"], data[f"{name}|labels"], data[f"{name}|meta"]
    cdir = str(tmp_path) + "/"
    wa.build_cache_shard(1, int(metric), X, labels, 1000, 2, 0.5, wa.BuildParams(int(R), int(L), 1.0, cdir), 0, 1, 4)
    assert os.listdir(cdir) == [data[f"{name}|file_name"].tobytes().decode()]
    assert open(cdir + os.listdir(cdir)[0], "rb").read() == data[f"{name}|file"].tobytes()


def test_integer_valued_graph_equals_the_references_with_its_tie_order(oracle, wa, tmp_path, monkeypatch):
    """Integer-valued vectors (SIFT-like): prunes and neighbour sorts are full of exactly equidistant candidates, and
    the reference orders them by libstdc++'s std::sort on distance alone (vamana/index.h:77-78, graph.h:106).  In
    reference-tie-order mode the oracle's builder and the product's host builder reproduce the file the REAL
    reference wrote; in the default (ties by id) mode they agree with each other and differ from it."""
    import golden_util as gu
    data = gu.load_build()
    name = "int_l2"
    X, labels, (R, L, metric) = data[f"{name}|X"], data[f"{name}|labels"], data[f"{name}|meta"]
    want = data[f"{name}|file"].tobytes()
    n, d = X.shape
    for mode in ("1", "0"):
        monkeypatch.setenv("WANN_REF_TIES", mode)
        monkeypatch.setenv("ORC_REF_TIES", mode)
        cdir = str(tmp_path / ("m" + mode)) + "/"
        os.makedirs(cdir)
        wa.build_cache_shard(1, int(metric), X, labels, 1000, 2, 0.5, wa.BuildParams(int(R), int(L), 1.0, cdir), 0, 1, 3)
        got = open(cdir + os.listdir(cdir)[0], "rb").read()
        rows = oracle.vamana_build(oracle.pad_rows(X), d, int(metric), 0, n, int(R), int(L), 1.0)
        opath = str(tmp_path / ("oracle" + mode + ".bin"))
        oracle.graph_save(opath, rows)
        assert open(opath, "rb").read() == got, f"mode {mode}: host builder and oracle builder disagree"
        assert (got == want) == (mode == "1"), f"mode {mode}"


def test_sharded_build_equals_whole_build(wa, tmp_path):
    n, d = 2500, 32
    X = sift_like(n, d, 3)(n)
    labels = distinct_labels(n, 9)
    whole, parts = str(tmp_path / "whole") + "/", str(tmp_path / "parts") + "/"
    os.makedirs(whole), os.makedirs(parts)
    bp = lambda c: wa.BuildParams(16, 32, 1.0, c)  # noqa: E731
    wa.build_cache_shard(3, 0, X, labels, 300, 2, 0.5, bp(whole), 0, 1, 2)
    for s in range(3):
        wa.build_cache_shard(3, 0, X, labels, 300, 2, 0.5, bp(parts), s, 3, 2)
    assert _read(whole) == _read(parts)
    # thread count does not change the graphs
    again = str(tmp_path / "again") + "/"
    os.makedirs(again)
    wa.build_cache_shard(3, 0, X, labels, 300, 2, 0.5, bp(again), 0, 1, 7)
    assert _read(whole) == _read(again)


def test_graph_quality_by_oracle_recall(oracle, wa, tmp_path):
    """The builder's graphs must give the oracle's search a sane recall against exact ground truth."""
    from util import brute_force_gt, recall, windows
    n, d, nq = 4000, 64, 200
    g = sift_like(n, d, 5)
    X, Q = g(n), g(nq)
    labels = distinct_labels(n, 2)
    cache = str(tmp_path) + "/"
    wa.build_cache_shard(3, 0, X, labels, 500, 2, 0.5, wa.BuildParams(32, 64, 1.0, cache), 0, 1, 4)
    idx = oracle.VamanaRangeFilterTreeIndexFloatEuclidian(X, labels, cutoff=500, split_factor=2, build_params=oracle.BuildParams(32, 64, 1.0, cache))
    for p in (-4, -2, 0):
        W = windows(labels, nq, p, 11)
        ids, _ = idx.batch_search(Q, W, nq, "optimized_postfilter", oracle.QueryParams(10, 40, final_beam_multiply=2))
        gt = brute_force_gt(X, labels, Q, W, 10, "l2")
        assert recall(gt, ids) > 0.9


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 8, 16, 17, 33, 64])
def test_tiny_partitions_build_like_the_oracle(oracle, wa, tmp_path, n):
    """Partitions so small that the batch cap equals their size (empty doubling rounds, single-point batches)."""
    rng = np.random.default_rng(n)
    X = rng.integers(0, 100, size=(n, 6)).astype(np.float32)
    labels = distinct_labels(n, 3)
    pdir, odir = str(tmp_path / "p") + "/", str(tmp_path / "o") + "/"
    os.makedirs(pdir), os.makedirs(odir)
    wa.build_cache_shard(1, 0, X, labels, 1000, 2, 0.5, wa.BuildParams(4, 8, 1.0, pdir), 0, 1, 2)
    oracle.PostfilterVamanaIndexFloatEuclidian(X, filters=labels, build_params=oracle.BuildParams(4, 8, 1.0, odir), threads=2)
    assert _read(pdir) == _read(odir)
