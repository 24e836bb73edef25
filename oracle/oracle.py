"""oracle/oracle.py -- TEST INFRASTRUCTURE ONLY.

ctypes front end of the CPU restatement (oracle/restate.cpp -> oracle/liboracle.so) plus a loader
for the REAL reference module when oracle/_ref holds a build of it (`make -C oracle ref`).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module, and
only as the checker.  The product (rangefilteredann_amd/) never does.

The Python classes mirror the reference's pybind11 surface (python_bindings/python_bindings.cpp:
111-157, 204-213) so a test can drive the oracle, the real reference and the product with the
same lines.
"""
from __future__ import annotations

import ctypes as C
import glob
import importlib.machinery
import importlib.util
import os
import subprocess
import sys
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

L2, MIPS = 0, 1
INTEGER = 2  # | : integer-valued rows of a uint8 / int8 point set (int32 distances, restate.h ORC_INTEGER)
PREFILTER, POSTFILTER, TREE_PREFILTER, TREE_VAMANA, SUPER = range(5)


def build(force: bool = False) -> str:
    """Compile the restatement with gcc (seconds)."""
    src = os.path.join(_HERE, "restate.cpp")
    hdr = os.path.join(_HERE, "restate.h")
    if (force or not os.path.exists(_LIB_PATH)
            or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(hdr))):
        subprocess.check_call(["make", "-C", _HERE, "restate"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class _QP(C.Structure):
    _fields_ = [("k", C.c_int64), ("beam", C.c_int64), ("limit", C.c_int64),
                ("degree_limit", C.c_int64), ("final_beam_multiply", C.c_int64),
                ("max_beam", C.c_int64), ("cut", C.c_double), ("has_ratio", C.c_int32),
                ("ratio", C.c_float), ("verbose", C.c_int32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.orc_random_permutation.restype = None
        L.orc_random_permutation.argtypes = [C.c_int64, C.c_void_p]
        L.orc_hash64_2.restype = C.c_uint64
        L.orc_hash64_2.argtypes = [C.c_uint64]
        L.orc_hash_bits.restype = C.c_int
        L.orc_hash_bits.argtypes = [C.c_int64]
        L.orc_distance.restype = C.c_float
        L.orc_distance.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint32]
        L.orc_last_error.restype = C.c_char_p
        L.orc_index_create.restype = C.c_void_p
        L.orc_index_create.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_int64,
                                       C.c_void_p, C.c_int32, C.c_double, C.c_double, C.c_int64,
                                       C.c_int64, C.c_double, C.c_char_p, C.c_int]
        L.orc_index_destroy.argtypes = [C.c_void_p]
        L.orc_batch_search.restype = C.c_int
        L.orc_batch_search.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_char_p,
                                       C.POINTER(_QP), C.c_void_p, C.c_void_p, C.c_int,
                                       C.c_void_p]
        L.orc_beam_search.restype = C.c_int64
        L.orc_beam_search.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64,
                                      C.c_int64, C.c_int, C.c_int64, C.c_void_p, C.c_int64,
                                      C.c_int64, C.c_int64, C.c_int64, C.c_double, C.c_int64,
                                      C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p]
        L.orc_graph_load.restype = C.c_int
        L.orc_graph_load.argtypes = [C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64),
                                     C.POINTER(C.c_int64)]
        L.orc_graph_save.restype = C.c_int
        L.orc_graph_save.argtypes = [C.c_char_p, C.c_void_p, C.c_int64, C.c_int64]
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_vamana_build.restype = C.c_int
        L.orc_vamana_build.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int64,
                                       C.c_int64, C.c_int64, C.c_int64, C.c_double, C.c_void_p,
                                       C.c_int]
        L.orc_num_levels.restype = C.c_int64
        L.orc_num_levels.argtypes = [C.c_void_p]
        L.orc_level_size.restype = C.c_int64
        L.orc_level_size.argtypes = [C.c_void_p, C.c_int64]
        L.orc_partition_range.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_int64),
                                          C.POINTER(C.c_int64)]
        L.orc_partition_graph.restype = C.c_void_p
        L.orc_partition_graph.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_int64),
                                          C.POINTER(C.c_int64)]
        L.orc_decoding.restype = C.c_void_p
        L.orc_decoding.argtypes = [C.c_void_p]
        L.orc_sorted_labels.restype = C.c_void_p
        L.orc_sorted_labels.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def default_threads() -> int:
    env = os.environ.get("PARLAY_NUM_THREADS")  # run_our_method.py:131
    return int(env) if env else (os.cpu_count() or 1)


# ----------------------------------------------------------------------------- params
class BuildParams:
    """BuildParams(max_degree, limit, alpha, cache_path) -- python_bindings.cpp:211-213."""

    def __init__(self, max_degree=64, limit=500, alpha=1.175, cache_path="index_cache"):
        self.R, self.L, self.alpha, self.cache_path = int(max_degree), int(limit), float(alpha), str(cache_path)


class QueryParams:
    """QueryParams(k, beam_width, cut, limit, degree_limit, final_beam_multiply,
    postfiltering_max_beam, min_query_to_bucket_ratio, verbose) -- python_bindings.cpp:204-209."""

    def __init__(self, k, beam_width, cut=1.35, limit=10_000_000, degree_limit=10_000,
                 final_beam_multiply=1, postfiltering_max_beam=10000,
                 min_query_to_bucket_ratio=None, verbose=False):
        self.c = _QP(int(k), int(beam_width), int(limit), int(degree_limit),
                     int(final_beam_multiply), int(postfiltering_max_beam), float(cut),
                     0 if min_query_to_bucket_ratio is None else 1,
                     0.0 if min_query_to_bucket_ratio is None else float(min_query_to_bucket_ratio),
                     int(bool(verbose)))
        self.k = int(k)


# ----------------------------------------------------------------------------- indexes
class _Index:
    KIND = None

    def __init__(self, metric, points, labels, cutoff=1000, split_factor=2, shift_factor=0.5,
                 build_params: Optional[BuildParams] = None, threads: Optional[int] = None):
        bp = build_params or BuildParams()
        pts = np.ascontiguousarray(points, dtype=np.float32)
        lab = np.ascontiguousarray(labels, dtype=np.float32)
        if pts.ndim != 2:
            raise RuntimeError("points numpy array must be 2-dimensional")
        if lab.ndim != 1:
            raise RuntimeError("filter data numpy array must be 1-dimensional")
        if lab.shape[0] != pts.shape[0]:
            raise RuntimeError("filter data numpy array must have the same number of elements as the points array")
        self.n, self.d = pts.shape
        self.threads = threads or default_threads()
        self.h = lib().orc_index_create(self.KIND, metric, _ptr(pts), self.n, self.d, _ptr(lab),
                                        int(cutoff), float(split_factor), float(shift_factor),
                                        bp.R, bp.L, bp.alpha, bp.cache_path.encode(), self.threads)
        if not self.h:
            raise RuntimeError(lib().orc_last_error().decode())
        self.last_counters = None

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_index_destroy(self.h)
            self.h = None

    def _search(self, queries, filters, num_queries, method, qp: QueryParams):
        q = np.ascontiguousarray(queries, dtype=np.float32)
        r = np.ascontiguousarray(np.asarray(filters, dtype=np.float64).astype(np.float32))
        nq = int(num_queries)
        ids = np.empty((nq, qp.k), dtype=np.uint32)
        dists = np.empty((nq, qp.k), dtype=np.float32)
        ctr = np.zeros(3, dtype=np.int64)
        rc = lib().orc_batch_search(self.h, _ptr(q), _ptr(r), nq, method.encode(), C.byref(qp.c),
                                    _ptr(ids), _ptr(dists), self.threads, _ptr(ctr))
        if rc:
            raise RuntimeError(lib().orc_last_error().decode())
        self.last_counters = {"searches": int(ctr[0]), "hops": int(ctr[1]), "dist_cmps": int(ctr[2])}
        return ids, dists

    # introspection
    def levels(self):
        return [int(lib().orc_level_size(self.h, l)) for l in range(lib().orc_num_levels(self.h))]

    def partition_range(self, level, idx):
        s, e = C.c_int64(), C.c_int64()
        lib().orc_partition_range(self.h, level, idx, C.byref(s), C.byref(e))
        return s.value, e.value

    def partition_graph(self, level, idx):
        n, md = C.c_int64(), C.c_int64()
        p = lib().orc_partition_graph(self.h, level, idx, C.byref(n), C.byref(md))
        arr = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_int32)), shape=(n.value, md.value + 1))
        return arr.copy()

    def decoding(self):
        p = lib().orc_decoding(self.h)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_int64)), shape=(self.n,)).copy()

    def sorted_labels(self):
        p = lib().orc_sorted_labels(self.h)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(self.n,)).copy()


def _mk(kind, metric, tree_method, elem=None):
    """elem = np.uint8 / np.int8: the reference's byte variants (euclidian_point.h:44-60, mips_point.h:44-58
    accumulate in int32 and cast to float).  Points and queries are cast to the element type as pybind's
    py::array_t<T> does; the rows then hold the integer values and the restatement accumulates in int32 too
    (metric | INTEGER), exact for any dimension."""
    def cast(a):
        if elem is None:
            return a
        return np.asarray(a).astype(elem).astype(np.float32)

    class K(_Index):
        KIND = kind

        def __init__(self, points, filter_values=None, cutoff=1000, split_factor=2,
                     shift_factor=0.5, build_params=None, filters=None, threads=None):
            lab = filter_values if filter_values is not None else filters
            super().__init__(metric | (INTEGER if elem is not None else 0), cast(points), lab, cutoff, split_factor, shift_factor,
                             build_params, threads)

        if tree_method:
            def batch_search(self, queries, filters, num_queries, query_method, query_params):
                return self._search(cast(queries), filters, num_queries, query_method, query_params)
        else:
            def batch_search(self, queries, filters, num_queries, query_params):
                return self._search(cast(queries), filters, num_queries, "", query_params)
    return K


PrefilterIndexFloatEuclidian = _mk(PREFILTER, L2, False)
PrefilterIndexFloatMips = _mk(PREFILTER, MIPS, False)
PostfilterVamanaIndexFloatEuclidian = _mk(POSTFILTER, L2, False)
PostfilterVamanaIndexFloatMips = _mk(POSTFILTER, MIPS, False)
RangeFilterTreeIndexFloatEuclidian = _mk(TREE_PREFILTER, L2, True)
RangeFilterTreeIndexFloatMips = _mk(TREE_PREFILTER, MIPS, True)
VamanaRangeFilterTreeIndexFloatEuclidian = _mk(TREE_VAMANA, L2, True)
VamanaRangeFilterTreeIndexFloatMips = _mk(TREE_VAMANA, MIPS, True)
SuperOptimizedPostfilterTreeIndexFloatEuclidian = _mk(SUPER, L2, False)
SuperOptimizedPostfilterTreeIndexFloatMips = _mk(SUPER, MIPS, False)
for _en, _et in (("UInt8", np.uint8), ("Int8", np.int8)):
    for _mn, _mc in (("Euclidian", L2), ("Mips", MIPS)):
        globals()[f"PrefilterIndex{_en}{_mn}"] = _mk(PREFILTER, _mc, False, _et)
        globals()[f"PostfilterVamanaIndex{_en}{_mn}"] = _mk(POSTFILTER, _mc, False, _et)
        globals()[f"RangeFilterTreeIndex{_en}{_mn}"] = _mk(TREE_PREFILTER, _mc, True, _et)
        globals()[f"VamanaRangeFilterTreeIndex{_en}{_mn}"] = _mk(TREE_VAMANA, _mc, True, _et)
        globals()[f"SuperOptimizedPostfilterTreeIndex{_en}{_mn}"] = _mk(SUPER, _mc, False, _et)


# ----------------------------------------------------------------------------- raw pieces
def hash64_2(x: int) -> int:
    return int(lib().orc_hash64_2(C.c_uint64(x & 0xFFFFFFFFFFFFFFFF)))


def random_permutation(n: int) -> np.ndarray:
    """parlay::random_permutation<int>(n) with the default generator (the builder's insertion order)."""
    out = np.empty(n, dtype=np.int32)
    lib().orc_random_permutation(n, out.ctypes.data_as(C.c_void_p))
    return out


def hash_bits(beam: int) -> int:
    return int(lib().orc_hash_bits(beam))


def distance(metric: int, p: np.ndarray, q: np.ndarray) -> float:
    p = np.ascontiguousarray(p, dtype=np.float32)
    q = np.ascontiguousarray(q, dtype=np.float32)
    return float(lib().orc_distance(metric, _ptr(p), _ptr(q), p.shape[0]))


def pad_rows(points: np.ndarray) -> np.ndarray:
    """Row-major float32 with rows zero padded to 64 bytes (point_range.h:39-44)."""
    pts = np.ascontiguousarray(points, dtype=np.float32)
    n, d = pts.shape
    stride = ((d * 4 + 63) // 64) * 16
    out = np.zeros((n, stride), dtype=np.float32)
    out[:, :d] = pts
    return out


def beam_search(graph_rows: np.ndarray, points_padded: np.ndarray, d: int, metric: int,
                subset_start: int, query: np.ndarray, query_id: int, beam: int, k: Optional[int] = None,
                cut: float = 1.35, limit: int = 10_000_000, degree_limit: int = 10_000, start_node: int = 0):
    """Raw beam search; graph_rows = (n, maxdeg+1) int32 with the degree in column 0."""
    g = np.ascontiguousarray(graph_rows, dtype=np.int32)
    n, md1 = g.shape
    q = np.ascontiguousarray(query, dtype=np.float32)
    ids = np.empty(beam, dtype=np.int32)
    dists = np.empty(beam, dtype=np.float32)
    vis_cap = min(int(limit), n) + 1
    vids = np.empty(vis_cap, dtype=np.int32)
    vd = np.empty(vis_cap, dtype=np.float32)
    nv, dc = C.c_int64(), C.c_int64()
    nb = lib().orc_beam_search(_ptr(g), n, md1 - 1, _ptr(points_padded), points_padded.shape[1], d,
                               metric, subset_start, _ptr(q), query_id, start_node,
                               beam if k is None else k, beam, cut, limit, degree_limit,
                               _ptr(ids), _ptr(dists), _ptr(vids), _ptr(vd), C.byref(nv), C.byref(dc))
    return ids[:nb].copy(), dists[:nb].copy(), vids[:nv.value].copy(), vd[:nv.value].copy(), dc.value


def graph_load(path: str) -> np.ndarray:
    rows, n, md = C.c_void_p(), C.c_int64(), C.c_int64()
    if lib().orc_graph_load(path.encode(), C.byref(rows), C.byref(n), C.byref(md)):
        raise IOError("cannot read graph " + path)
    arr = np.ctypeslib.as_array(C.cast(rows, C.POINTER(C.c_int32)), shape=(n.value, md.value + 1)).copy()
    lib().orc_free(rows)
    return arr


def graph_save(path: str, rows: np.ndarray) -> None:
    g = np.ascontiguousarray(rows, dtype=np.int32)
    if lib().orc_graph_save(path.encode(), _ptr(g), g.shape[0], g.shape[1] - 1):
        raise IOError("cannot write graph " + path)


def vamana_build(points_padded: np.ndarray, d: int, metric: int, subset_start: int, n: int,
                 R: int, L: int, alpha: float, threads: Optional[int] = None) -> np.ndarray:
    rows = np.zeros((n, R + 1), dtype=np.int32)
    if lib().orc_vamana_build(_ptr(points_padded), points_padded.shape[1], d, metric, subset_start,
                              n, R, L, alpha, _ptr(rows), threads or default_threads()):
        raise RuntimeError(lib().orc_last_error().decode())
    return rows


# ----------------------------------------------------------------------------- real reference
def reference_so(prefer=("native", "x86-64-v4", "x86-64-v3")) -> Optional[str]:
    """Path of a built reference module under oracle/_ref, or None."""
    for sub in prefer:
        hits = glob.glob(os.path.join(_HERE, "_ref", sub, "window_ann*.so"))
        if hits:
            return hits[0]
    return None


def load_reference(prefer=("native", "x86-64-v4", "x86-64-v3")):
    """Import the REAL reference's pybind11 module (oracle/_ref/<march>/window_ann*.so).

    Returns the module or None when no build is present.  The module's init symbol is
    PyInit_window_ann, so it is loaded under that name without touching sys.modules' entry for
    the product's own `window_ann` package: callers get the module object only."""
    path = reference_so(prefer)
    if path is None:
        return None
    loader = importlib.machinery.ExtensionFileLoader("window_ann", path)
    spec = importlib.util.spec_from_file_location("window_ann", path, loader=loader)
    mod = importlib.util.module_from_spec(spec)
    loader.exec_module(mod)
    return mod
