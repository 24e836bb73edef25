"""The unfiltered VamanaIndex API (python_bindings.cpp:92-109; ParlayANN/python/vamana_index.cpp, builder.cpp).
Golden vectors come from the real reference (tests/golden/make_vamana_golden.py).  CPU: the oracle's beam search with the
k / cut step (beamSearch.h:159-167) reproduces them from the reference-built graph files.  GPU: the product's classes
(file in, file out) return the same rows, and build_vamana_*_index writes the reference's graph file byte for byte."""
import os

import numpy as np
import pytest

import golden_util as gu
from util import REPO  # noqa: F401

CASES = {"l2": (0, "float_euclidian", "VamanaFloatEuclidianIndex"), "mips": (1, "float_mips", "VamanaFloatMipsIndex"),
         "u8": (0, "uint8_euclidian", "VamanaUInt8EuclidianIndex")}


def _golden():
    return np.load(os.path.join(gu.GOLDEN, "vamana_golden.npz"))


def _settings(g, name):
    for f in g.files:
        if f.startswith(name + "/ids|"):
            _, knn, beam = f.split("|")
            yield int(knn), int(beam)


def _write_bin(path, X):
    with open(path, "wb") as f:
        np.array(X.shape, dtype=np.uint32).tofile(f)
        X.tofile(f)


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_reproduces_reference_vamana_search(oracle, tmp_path, name):
    g = _golden()
    metric = CASES[name][0]
    X, Q = g[name + "/X"], g[name + "/Q"]
    n, d = X.shape
    gpath = tmp_path / "graph.bin"
    gpath.write_bytes(g[name + "/graph"].tobytes())
    rows = oracle.graph_load(str(gpath))
    Xp = oracle.pad_rows(X.astype(np.float32))
    n_set = 0
    for knn, beam in _settings(g, name):
        for i in range(Q.shape[0]):
            ids, dists, _, _, _ = oracle.beam_search(rows, Xp, d, metric, 0, Q[i].astype(np.float32), i, beam, k=knn, cut=1.35,
                                                     limit=n, degree_limit=rows.shape[1] - 1)
            assert len(ids) >= knn
            assert np.array_equal(ids[:knn].astype(np.uint32), g[f"{name}/ids|{knn}|{beam}"][i]), (name, knn, beam, i)
            assert np.array_equal(dists[:knn], g[f"{name}/dists|{knn}|{beam}"][i]), (name, knn, beam, i)
        n_set += 1
    assert n_set >= 4


def test_python_surface_has_the_vamana_variants(wa):
    for lower, cls in (("float_euclidian", "VamanaFloatEuclidianIndex"), ("float_mips", "VamanaFloatMipsIndex"),
                       ("uint8_euclidian", "VamanaUInt8EuclidianIndex"), ("uint8_mips", "VamanaUInt8MipsIndex"),
                       ("int8_euclidian", "VamanaInt8EuclidianIndex"), ("int8_mips", "VamanaInt8MipsIndex")):
        assert hasattr(wa, "build_vamana_" + lower + "_index") and hasattr(wa, cls)
    with pytest.raises(RuntimeError):
        wa.VamanaFloatEuclidianIndex("/nonexistent/points.bin", "/nonexistent/graph.bin", 1, 1)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(CASES))
def test_vamana_index_matches_reference(wa, gpu, tmp_path, name):
    g = _golden()
    metric, lower, cls = CASES[name]
    X, Q = g[name + "/X"], g[name + "/Q"]
    n, d = X.shape
    nq = Q.shape[0]
    data, graph = str(tmp_path / "points.bin"), str(tmp_path / "graph.bin")
    _write_bin(data, X)
    with open(graph, "wb") as f:
        f.write(g[name + "/graph"].tobytes())
    # the binding calls its first argument "index_path" but it is the POINT file (vamana_index.cpp:46 vs python_bindings.cpp:98-100)
    idx = getattr(wa, cls)(data, graph, n, d)
    for knn, beam in _settings(g, name):
        ids, dists = idx.batch_search(Q, nq, knn, beam)
        assert ids.dtype == np.uint32 and ids.shape == (nq, knn)
        assert np.array_equal(ids, g[f"{name}/ids|{knn}|{beam}"]), (name, knn, beam)
        assert np.array_equal(dists, g[f"{name}/dists|{knn}|{beam}"]), (name, knn, beam)
    qfile = str(tmp_path / "queries.bin")
    _write_bin(qfile, Q)
    ids2, dists2 = idx.batch_search_from_string(qfile, nq, 10, 40)
    assert np.array_equal(ids2, g[f"{name}/ids|10|40"]) and np.array_equal(dists2, g[f"{name}/dists|10|40"])
    # keyword call: the reference's (swapped) argument names
    idx_kw = getattr(wa, cls)(index_path=data, data_path=graph, num_points=n, dimensions=d)
    ids3, _ = idx_kw.batch_search(Q, nq, 10, 40)
    assert np.array_equal(ids3, ids2)
    # check_recall against a ground-truth file in the reference's format (types.h:33-74)
    Xf, Qf = X.astype(np.float64), Q.astype(np.float64)
    dm = -(Qf @ Xf.T) if metric == 1 else ((Qf[:, None, :] - Xf[None, :, :]) ** 2).sum(2)
    order = np.argsort(dm, axis=1, kind="stable")[:, :20]
    gt = str(tmp_path / "gt.bin")
    with open(gt, "wb") as f:
        np.array([nq, 20], dtype=np.int32).tofile(f)
        order.astype(np.uint32).tofile(f)
        np.take_along_axis(dm, order, 1).astype(np.float32).tofile(f)
    rec = idx.check_recall(gt, ids2, 10)
    assert 0.5 < rec <= 1.0001


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["mips"])
def test_build_vamana_writes_the_references_graph_file(wa, gpu, tmp_path, name):
    """continuous coordinates (no distance ties): the GPU builder's file equals the reference builder's byte for byte"""
    g = _golden()
    metric, lower, cls = CASES[name]
    R, L, a1000 = [int(x) for x in g[name + "/meta"]]
    data, graph = str(tmp_path / "points.bin"), str(tmp_path / "graph.bin")
    _write_bin(data, g[name + "/X"])
    getattr(wa, "build_vamana_" + lower + "_index")("ignored", data, graph, R, L, a1000 / 1000.0)
    assert open(graph, "rb").read() == g[name + "/graph"].tobytes()
