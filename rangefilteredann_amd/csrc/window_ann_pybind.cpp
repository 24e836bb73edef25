// window_ann_pybind.cpp -- pybind11 shim over the C ABI (include/wann.h) that re-exports the
// Python surface of the reference's `window_ann` module for the window-search path
// (python_bindings/python_bindings.cpp:111-157,204-213 of JoshEngels/RangeFilteredANN):
// same class names, constructor keywords, batch_search signatures and return tuple, so
// experiments/wrapper.py and run_our_method.py drive this engine unchanged.
//
// Everything computes on the GPU through libwann.so; without a gfx950 device constructors raise.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <optional>
#include <stdexcept>
#include <string>
#include <vector>
#include <cstdio>

#include "../../include/wann.h"

namespace py = pybind11;
using namespace pybind11::literals;

namespace wannpy {

struct BuildParams {  // python_bindings.cpp:211-213
  long max_degree, limit;
  double alpha;
  std::string cache_path;
  BuildParams(long R, long L, double a, std::string c) : max_degree(R), limit(L), alpha(a), cache_path(std::move(c)) {}
};

struct QueryParams {  // python_bindings.cpp:204-209
  wann_query_params c{};
  QueryParams(long k, long beam, double cut, long limit, long degree_limit, long final_beam_multiply,
              long postfiltering_max_beam, std::optional<float> ratio, bool verbose) {
    c.k = k;
    c.beam_width = beam;
    c.cut = cut;
    c.limit = limit;
    c.degree_limit = degree_limit;
    c.final_beam_multiply = final_beam_multiply;
    c.postfiltering_max_beam = postfiltering_max_beam;
    c.has_min_query_to_bucket_ratio = ratio.has_value();
    c.min_query_to_bucket_ratio = ratio.value_or(0.f);
    c.verbose = verbose;
  }
};

static const BuildParams &default_build_params() {  // python_bindings.cpp:88
  static BuildParams bp(64, 500, 1.175, "index_cache");
  return bp;
}

[[noreturn]] static void raise_last(const char *what) {
  throw std::runtime_error(std::string(what) + ": " + wann_last_error());
}

using FArray = py::array_t<float, py::array::c_style | py::array::forcecast>;
using NeighborsAndDistances = std::pair<py::array_t<unsigned int>, py::array_t<float>>;

class Index {
 public:
  Index(int kind, int metric, int dtype, py::array points, FArray labels, int32_t cutoff, double split,
        double shift, const BuildParams &bp)
      : kind_(kind), dtype_(dtype) {
    // py::array_t<T> semantics of the reference (python_bindings.cpp:113,121,...): any array is cast to T
    py::array pts = dtype == WANN_DTYPE_F32 ? py::array(FArray::ensure(points))
                  : dtype == WANN_DTYPE_U8 ? py::array(py::array_t<uint8_t, py::array::c_style | py::array::forcecast>::ensure(points))
                                           : py::array(py::array_t<int8_t, py::array::c_style | py::array::forcecast>::ensure(points));
    if (!pts) throw std::runtime_error("points must be convertible to an array of the index's element type");
    if (pts.ndim() != 2) throw std::runtime_error("points numpy array must be 2-dimensional");           // tree_utils.h:46
    if (labels.ndim() != 1) throw std::runtime_error("filter data numpy array must be 1-dimensional");    // tree_utils.h:53
    if (labels.shape(0) != pts.shape(0))
      throw std::runtime_error("filter data numpy array must have the same number of elements as the points array");
    wann_build_params wb{bp.max_degree, bp.limit, bp.alpha, bp.cache_path.c_str()};
    const void *pp = pts.data();
    const float *lp = labels.data();
    int64_t n = pts.shape(0), d = pts.shape(1);
    {
      py::gil_scoped_release nogil;
      h_ = wann_index_create(kind, metric, dtype, pp, n, d, lp, cutoff, split, shift, &wb, device_from_env(), 0);
    }
    if (!h_) raise_last("index construction failed");
  }
  ~Index() { wann_index_destroy(h_); }
  Index(const Index &) = delete;
  Index &operator=(const Index &) = delete;

  static int device_from_env() {
    // one process per GPU: LOCAL_RANK selects the device unless WANN_DEVICE overrides it
    if (const char *e = getenv("WANN_DEVICE")) return atoi(e);
    int ndev = wann_device_count();
    if (const char *e = getenv("LOCAL_RANK")) return ndev > 0 ? atoi(e) % ndev : 0;
    return 0;
  }

  NeighborsAndDistances search(py::array queries_in, py::object filters, uint64_t nq, const std::string &method,
                               const QueryParams &qp) {
    FArray fr = FArray::ensure(filters);
    if (!fr) throw std::runtime_error("filters must be a sequence of (lo, hi) pairs");
    if (fr.ndim() != 2 || fr.shape(1) != 2 || (uint64_t)fr.shape(0) < nq)
      throw std::runtime_error("filters must have shape (num_queries, 2)");
    py::array queries = dtype_ == WANN_DTYPE_F32 ? py::array(FArray::ensure(queries_in))
                      : dtype_ == WANN_DTYPE_U8 ? py::array(py::array_t<uint8_t, py::array::c_style | py::array::forcecast>::ensure(queries_in))
                                                : py::array(py::array_t<int8_t, py::array::c_style | py::array::forcecast>::ensure(queries_in));
    if (!queries) throw std::runtime_error("queries must be convertible to an array of the index's element type");
    if (queries.ndim() != 2 || (uint64_t)queries.shape(0) < nq || queries.shape(1) != wann_dim(h_))
      throw std::runtime_error("queries must have shape (num_queries, dimension)");
    size_t k = (size_t)qp.c.k;
    py::array_t<unsigned int> ids({(size_t)nq, k});
    py::array_t<float> dists({(size_t)nq, k});
    const void *qptr = queries.data();
    const float *rptr = fr.data();
    unsigned int *ip = ids.mutable_data();
    float *dp = dists.mutable_data();
    int rc;
    {
      py::gil_scoped_release nogil;
      rc = wann_batch_search(h_, qptr, rptr, (int64_t)nq, method.c_str(), &qp.c, ip, dp);
    }
    if (rc) raise_last("batch_search failed");
    return std::make_pair(ids, dists);
  }

  // device-resident call: every *_ptr is a device address on this index's GPU (e.g. torch .data_ptr())
  void search_device(uint64_t q_ptr, uint64_t r_ptr, int64_t nq, int64_t query_id_base, const std::string &method,
                     const QueryParams &qp, uint64_t ids_ptr, uint64_t dists_ptr, uint64_t stream) {
    int rc;
    {
      py::gil_scoped_release nogil;
      rc = wann_batch_search_device(h_, (const void *)q_ptr, (const float *)r_ptr, nq, query_id_base, method.c_str(),
                                    &qp.c, (uint32_t *)ids_ptr, (float *)dists_ptr, (void *)stream);
    }
    if (rc) raise_last("batch_search_device failed");
  }

  // ... for queries that are not a contiguous range of their batch: query_ids_ptr = device array of their global row numbers (int64)
  void search_device_ids(uint64_t q_ptr, uint64_t r_ptr, int64_t nq, uint64_t qids_ptr, const std::string &method, const QueryParams &qp,
                         uint64_t ids_ptr, uint64_t dists_ptr, uint64_t stream) {
    int rc;
    {
      py::gil_scoped_release nogil;
      rc = wann_batch_search_device_ids(h_, (const void *)q_ptr, (const float *)r_ptr, nq, (const int64_t *)qids_ptr, method.c_str(), &qp.c,
                                        (uint32_t *)ids_ptr, (float *)dists_ptr, (void *)stream);
    }
    if (rc) raise_last("batch_search_device_ids failed");
  }

  // asynchronous device-resident call: returns a ticket at once; wait(ticket) blocks until that batch's rows are in place
  int64_t search_device_async(uint64_t q_ptr, uint64_t r_ptr, int64_t nq, int64_t query_id_base, const std::string &method,
                              const QueryParams &qp, uint64_t ids_ptr, uint64_t dists_ptr, uint64_t after_stream) {
    int64_t ticket = -1;
    int rc;
    {
      py::gil_scoped_release nogil;
      rc = wann_batch_search_device_async(h_, (const void *)q_ptr, (const float *)r_ptr, nq, query_id_base, method.c_str(), &qp.c,
                                          (uint32_t *)ids_ptr, (float *)dists_ptr, (void *)after_stream, &ticket);
    }
    if (rc) raise_last("batch_search_device_async failed");
    return ticket;
  }
  // predicted work per query (hops) of a batch of windows: what a cost-balanced shard cut balances
  py::array_t<float> predict_costs(py::array_t<float, py::array::c_style | py::array::forcecast> filters, const std::string &method,
                                   const QueryParams &qp) {
    if (filters.ndim() != 2 || filters.shape(1) != 2) throw std::runtime_error("filters must be (num_queries, 2)");
    const int64_t nq = filters.shape(0);
    py::array_t<float> out(nq);
    int rc;
    {
      py::gil_scoped_release nogil;
      rc = wann_predict_costs(h_, filters.data(), nq, method.c_str(), &qp.c, out.mutable_data());
    }
    if (rc) raise_last("predict_costs failed");
    return out;
  }
  py::dict wait(int64_t ticket) {
    wann_counters c;
    int rc;
    {
      py::gil_scoped_release nogil;
      rc = wann_wait(h_, ticket, &c);
    }
    if (rc) raise_last("wait failed");
    return counters_dict(c);
  }

  py::dict counters() const {
    wann_counters c;
    wann_get_counters(h_, &c);
    return counters_dict(c);
  }
  static py::dict counters_dict(const wann_counters &c) {
    py::dict d;
    d["beam_searches"] = c.beam_searches;
    d["hops"] = c.hops;
    d["dist_cmps"] = c.dist_cmps;
    d["brute_rows"] = c.brute_rows;
    d["label_reads"] = c.label_reads;
    d["rounds"] = c.rounds;
    d["spec_searches"] = c.spec_searches;
    d["spec_hops"] = c.spec_hops;
    d["spec_dist_cmps"] = c.spec_dist_cmps;
    d["gemm_queries"] = c.gemm_queries;
    d["gemm_unproven"] = c.gemm_unproven;
    d["gemm_rescued"] = c.gemm_rescued;
    d["deep_handoffs"] = c.deep_handoffs;
    d["lookaheads_used"] = c.lookaheads_used;
    d["big_searches"] = c.big_searches;
    d["big_hops"] = c.big_hops;
    d["packet_hops"] = c.packet_hops;
    d["own_scorings"] = c.own_scorings;
    d["prefetched_hops"] = c.prefetched_hops;
    d["poll_timeouts"] = c.poll_timeouts;
    d["lookaheads_issued"] = c.lookaheads_issued;
    d["recovered_continuations"] = c.recovered_continuations;
    d["device_ms"] = c.device_ms;
    d["search_kernel_ms"] = c.search_kernel_ms;
    return d;
  }
  std::vector<int64_t> levels() const {
    std::vector<int64_t> v;
    for (int64_t l = 0; l < wann_num_levels(h_); l++) v.push_back(wann_level_size(h_, l));
    return v;
  }
  std::pair<int64_t, int64_t> partition_range(int64_t level, int64_t idx) const {
    int64_t s, e;
    if (wann_partition_range(h_, level, idx, &s, &e)) raise_last("partition_range");
    return {s, e};
  }
  int64_t max_degree() const { return wann_max_degree(h_); }
  py::array_t<int32_t> partition_graph(int64_t level, int64_t idx, int64_t max_degree) const {
    auto [s, e] = partition_range(level, idx);
    py::array_t<int32_t> rows({(size_t)(e - s), (size_t)(max_degree + 1)});
    if (wann_partition_graph(h_, level, idx, rows.mutable_data(), e - s, max_degree)) raise_last("partition_graph");
    return rows;
  }
  int64_t device_bytes() const { return wann_device_bytes(h_); }
  int num_replicas() const { return wann_num_replicas(h_); }
  int64_t num_points() const { return wann_num_points(h_); }
  int64_t dim() const { return wann_dim(h_); }

 private:
  int kind_, dtype_;
  wann_index *h_ = nullptr;
};

template <int KIND, int METRIC, int DTYPE>
struct IndexT : Index {
  using Index::Index;
};

template <typename C>
static void common_defs(py::class_<C> &c) {
  c.def("batch_search_device", &C::search_device, "queries_ptr"_a, "filters_ptr"_a, "num_queries"_a,
        "query_id_base"_a, "query_method"_a, "query_params"_a, "ids_ptr"_a, "dists_ptr"_a, "stream"_a = 0)
      .def("batch_search_device_ids", &C::search_device_ids, "queries_ptr"_a, "filters_ptr"_a, "num_queries"_a, "query_ids_ptr"_a, "query_method"_a,
           "query_params"_a, "ids_ptr"_a, "dists_ptr"_a, "stream"_a = 0)
      .def("batch_search_device_async", &C::search_device_async, "queries_ptr"_a, "filters_ptr"_a, "num_queries"_a, "query_id_base"_a,
           "query_method"_a, "query_params"_a, "ids_ptr"_a, "dists_ptr"_a, "after_stream"_a = 0)
      .def("wait", &C::wait, "ticket"_a)
      .def("predict_costs", &C::predict_costs, "filters"_a, "query_method"_a, "query_params"_a)
      .def("counters", &C::counters)
      .def("levels", &C::levels)
      .def("partition_range", &C::partition_range)
      .def("partition_graph", &C::partition_graph, "level"_a, "idx"_a, "max_degree"_a)
      .def("device_bytes", &C::device_bytes)
      .def("max_degree", &C::max_degree)
      .def("num_replicas", &C::num_replicas)
      .def("num_points", &C::num_points)
      .def("dim", &C::dim);
}

template <int METRIC, int DTYPE>
static void add_variant(py::module_ &m, const std::string &agnostic) {
  {
    using C = IndexT<WANN_KIND_PREFILTER, METRIC, DTYPE>;
    py::class_<C> c(m, ("PrefilterIndex" + agnostic).c_str());
    c.def(py::init([](py::array points, FArray fv, const BuildParams &bp) {
            return new C(WANN_KIND_PREFILTER, METRIC, DTYPE, points, fv, 1000, 2, 0.5, bp);
          }),
          "points"_a, "filter_values"_a, "build_params"_a = default_build_params())
        .def("batch_search",
             [](C &self, py::array q, py::object f, uint64_t nq, const QueryParams &qp) { return self.search(q, f, nq, "", qp); },
             "queries"_a, "filters"_a, "num_queries"_a, "query_params"_a);
    common_defs(c);
  }
  {
    using C = IndexT<WANN_KIND_TREE_PREFILTER, METRIC, DTYPE>;
    py::class_<C> c(m, ("RangeFilterTreeIndex" + agnostic).c_str());
    c.def(py::init([](py::array points, FArray fv, int32_t cutoff, size_t split, const BuildParams &bp) {
            return new C(WANN_KIND_TREE_PREFILTER, METRIC, DTYPE, points, fv, cutoff, (double)split, 0.5, bp);
          }),
          "points"_a, "filter_values"_a, "cutoff"_a = 1000, "split_factor"_a = 2,
          "build_params"_a = default_build_params())
        .def("batch_search",
             [](C &self, py::array q, py::object f, uint64_t nq, const std::string &method, const QueryParams &qp) {
               return self.search(q, f, nq, method, qp);
             },
             "queries"_a, "filters"_a, "num_queries"_a, "query_method"_a, "query_params"_a);
    common_defs(c);
  }
  {
    using C = IndexT<WANN_KIND_POSTFILTER, METRIC, DTYPE>;
    py::class_<C> c(m, ("PostfilterVamanaIndex" + agnostic).c_str());
    c.def(py::init([](py::array points, FArray fv, const BuildParams &bp) {
            return new C(WANN_KIND_POSTFILTER, METRIC, DTYPE, points, fv, 1000, 2, 0.5, bp);
          }),
          "points"_a, "filters"_a, "build_params"_a = default_build_params())
        .def("batch_search",
             [](C &self, py::array q, py::object f, uint64_t nq, const QueryParams &qp) { return self.search(q, f, nq, "", qp); },
             "queries"_a, "filters"_a, "num_queries"_a, "query_params"_a);
    common_defs(c);
  }
  {
    using C = IndexT<WANN_KIND_TREE_VAMANA, METRIC, DTYPE>;
    py::class_<C> c(m, ("VamanaRangeFilterTreeIndex" + agnostic).c_str());
    c.def(py::init([](py::array points, FArray fv, int32_t cutoff, size_t split, const BuildParams &bp) {
            return new C(WANN_KIND_TREE_VAMANA, METRIC, DTYPE, points, fv, cutoff, (double)split, 0.5, bp);
          }),
          "points"_a, "filter_values"_a, "cutoff"_a = 1000, "split_factor"_a = 2,
          "build_params"_a = default_build_params())
        .def("batch_search",
             [](C &self, py::array q, py::object f, uint64_t nq, const std::string &method, const QueryParams &qp) {
               return self.search(q, f, nq, method, qp);
             },
             "queries"_a, "filters"_a, "num_queries"_a, "query_method"_a, "query_params"_a);
    common_defs(c);
  }
  {
    using C = IndexT<WANN_KIND_SUPER, METRIC, DTYPE>;
    py::class_<C> c(m, ("SuperOptimizedPostfilterTreeIndex" + agnostic).c_str());
    c.def(py::init([](py::array points, FArray fv, int32_t cutoff, float split, float shift, const BuildParams &bp) {
            return new C(WANN_KIND_SUPER, METRIC, DTYPE, points, fv, cutoff, (double)split, (double)shift, bp);
          }),
          "points"_a, "filter_values"_a, "cutoff"_a = 1000, "split_factor"_a = 2, "shift_factor"_a = 0.5,
          "build_params"_a = default_build_params())
        .def("batch_search",
             [](C &self, py::array q, py::object f, uint64_t nq, const QueryParams &qp) { return self.search(q, f, nq, "", qp); },
             "queries"_a, "filters"_a, "num_queries"_a, "query_params"_a);
    common_defs(c);
  }
}

// ---- unfiltered VamanaIndex<T,Point> (ParlayANN/python/vamana_index.cpp:42-76, python_bindings.cpp:92-109) -------------
// The reference's constructor is VamanaIndex(data_path, index_path, num_points, dimensions) but its binding names the
// arguments "index_path", "data_path" in that order: the FIRST argument (keyword index_path) is the point file and the
// second (keyword data_path) the graph file.  Kept as is: positional and keyword calls behave like the reference's.
template <int METRIC, int DTYPE>
struct VamanaIndexT {
  wann_vamana *h = nullptr;
  VamanaIndexT(const std::string &first_called_index_path, const std::string &second_called_data_path, size_t num_points, size_t dimensions) {
    h = wann_vamana_open(METRIC, DTYPE, first_called_index_path.c_str(), second_called_data_path.c_str(), 0);
    if (!h) raise_last("VamanaIndex");
    // (the reference asserts these; assert() is compiled out of its release build)
    (void)num_points;
    (void)dimensions;
  }
  ~VamanaIndexT() { wann_vamana_close(h); }
  VamanaIndexT(const VamanaIndexT &) = delete;
  NeighborsAndDistances search_raw(const void *q, uint64_t nq, uint64_t knn, uint64_t beam) {
    py::array_t<unsigned int> ids({(size_t)nq, (size_t)knn});
    py::array_t<float> dists({(size_t)nq, (size_t)knn});
    if (wann_vamana_batch_search(h, q, (int64_t)nq, (int64_t)knn, (int64_t)beam, ids.mutable_data(), dists.mutable_data()))
      raise_last("batch_search");
    return {std::move(ids), std::move(dists)};
  }
  NeighborsAndDistances batch_search(py::array queries, uint64_t nq, uint64_t knn, uint64_t beam) {
    const int64_t d = wann_vamana_dim(h);
    if (DTYPE == WANN_DTYPE_F32) {
      FArray q = FArray::ensure(queries);
      if (!q || q.ndim() != 2 || q.shape(1) != d || (uint64_t)q.shape(0) < nq) throw std::runtime_error("queries must be a (num_queries, dimensions) array");
      return search_raw(q.data(), nq, knn, beam);
    }
    if (DTYPE == WANN_DTYPE_U8) {
      auto q = py::array_t<uint8_t, py::array::c_style | py::array::forcecast>::ensure(queries);
      if (!q || q.ndim() != 2 || q.shape(1) != d || (uint64_t)q.shape(0) < nq) throw std::runtime_error("queries must be a (num_queries, dimensions) array");
      return search_raw(q.data(), nq, knn, beam);
    }
    auto q = py::array_t<int8_t, py::array::c_style | py::array::forcecast>::ensure(queries);
    if (!q || q.ndim() != 2 || q.shape(1) != d || (uint64_t)q.shape(0) < nq) throw std::runtime_error("queries must be a (num_queries, dimensions) array");
    return search_raw(q.data(), nq, knn, beam);
  }
  // queries from a point file (uint32 n, uint32 d, data)
  NeighborsAndDistances batch_search_from_string(const std::string &path, uint64_t nq, uint64_t knn, uint64_t beam) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open query file " + path);
    uint32_t head[2] = {0, 0};
    const size_t esz = DTYPE == WANN_DTYPE_F32 ? 4 : 1;
    std::vector<unsigned char> raw;
    bool ok = fread(head, 4, 2, f) == 2 && (int64_t)head[1] == wann_vamana_dim(h) && head[0] >= nq;
    if (ok) {
      raw.resize((size_t)nq * head[1] * esz);
      ok = raw.empty() || fread(raw.data(), 1, raw.size(), f) == raw.size();
    }
    fclose(f);
    if (!ok) throw std::runtime_error("query file does not hold num_queries points of the index's dimension: " + path);
    return search_raw(raw.data(), nq, knn, beam);
  }
  // ground truth file: int32 n, int32 width, n*width uint32 ids, n*width float distances (types.h:33-74); recall counts
  // every ground-truth point tied with the k-th distance (vamana_index.cpp:104-132) and is printed like the reference prints it
  double check_recall(const std::string &gfile, py::array_t<unsigned int, py::array::c_style | py::array::forcecast> neighbors, int k) {
    FILE *f = fopen(gfile.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open ground truth file " + gfile);
    int32_t head[2] = {0, 0};
    if (fread(head, 4, 2, f) != 2 || head[0] < 0 || head[1] < k) {
      fclose(f);
      throw std::runtime_error("bad ground truth file " + gfile);
    }
    const size_t n = (size_t)head[0], w = (size_t)head[1];
    std::vector<uint32_t> ids(n * w);
    std::vector<float> ds(n * w);
    const bool ok = fread(ids.data(), 4, n * w, f) == n * w && fread(ds.data(), 4, n * w, f) == n * w;
    fclose(f);
    if (!ok) throw std::runtime_error("truncated ground truth file " + gfile);
    if (neighbors.ndim() != 2 || (size_t)neighbors.shape(0) < n || neighbors.shape(1) < k) throw std::runtime_error("neighbors must be (n, >= k)");
    long correct = 0;
    for (size_t i = 0; i < n; i++) {
      size_t cnt = (size_t)k;
      while (cnt < w && ds[i * w + cnt] == ds[i * w + k - 1]) cnt++;
      for (size_t l = 0; l < cnt; l++)
        for (int j = 0; j < k; j++)
          if (neighbors.at(i, j) == ids[i * w + l]) {
            correct++;
            break;
          }
    }
    const double recall = (float)correct / (float)((size_t)k * n);
    py::print("Recall:", recall);
    return recall;
  }
};

template <int METRIC, int DTYPE>
static void add_vamana(py::module_ &m, const std::string &lower, const std::string &cls) {
  m.def(("build_vamana_" + lower + "_index").c_str(),
        [](const std::string &distance_metric, const std::string &data_file_path, const std::string &index_output_path, uint32_t graph_degree,
           uint32_t beam_width, float alpha) {
          (void)distance_metric;  // (unused by the reference too: the variant fixes the metric)
          if (wann_vamana_build_file(METRIC, DTYPE, data_file_path.c_str(), index_output_path.c_str(), graph_degree, beam_width, alpha, 0))
            raise_last("build_vamana_index");
        },
        "distance_metric"_a, "data_file_path"_a, "index_output_path"_a, "graph_degree"_a, "beam_width"_a, "alpha"_a);
  using V = VamanaIndexT<METRIC, DTYPE>;
  py::class_<V>(m, cls.c_str())
      .def(py::init<const std::string &, const std::string &, size_t, size_t>(), "index_path"_a, "data_path"_a, "num_points"_a, "dimensions"_a)
      .def("batch_search", &V::batch_search, "queries"_a, "num_queries"_a, "knn"_a, "beam_width"_a)
      .def("batch_search_from_string", &V::batch_search_from_string, "queries"_a, "num_queries"_a, "knn"_a, "beam_width"_a)
      .def("check_recall", &V::check_recall, "gFile"_a, "neighbors"_a, "k"_a);
}

}  // namespace wannpy

PYBIND11_MODULE(_window_ann, m) {
  using namespace wannpy;
  m.doc() = "WindowANN Python bindings -- MI355X (gfx950) engine";
  m.attr("__version__") = "mi355x-dev";

  py::module_ defaults = m.def_submodule("defaults");  // python_bindings.cpp:169-175
  defaults.attr("METRIC") = "Euclidian";
  defaults.attr("ALPHA") = 1.2;
  defaults.attr("GRAPH_DEGREE") = 64;
  defaults.attr("BEAMWIDTH") = 128;

  py::class_<QueryParams>(m, "QueryParams")
      .def(py::init<long, long, double, long, long, long, long, std::optional<float>, bool>(), "k"_a, "beam_width"_a,
           "cut"_a, "limit"_a, "degree_limit"_a, "final_beam_multiply"_a, "postfiltering_max_beam"_a,
           "min_query_to_bucket_ratio"_a, "verbose"_a);
  py::class_<BuildParams>(m, "BuildParams")
      .def(py::init<long, long, double, std::string>(), "max_degree"_a, "limit"_a, "alpha"_a, "cache_path"_a);

  add_variant<WANN_METRIC_L2, WANN_DTYPE_F32>(m, "FloatEuclidian");
  add_variant<WANN_METRIC_MIPS, WANN_DTYPE_F32>(m, "FloatMips");
  add_variant<WANN_METRIC_L2, WANN_DTYPE_U8>(m, "UInt8Euclidian");
  add_variant<WANN_METRIC_MIPS, WANN_DTYPE_U8>(m, "UInt8Mips");
  add_variant<WANN_METRIC_L2, WANN_DTYPE_I8>(m, "Int8Euclidian");
  add_variant<WANN_METRIC_MIPS, WANN_DTYPE_I8>(m, "Int8Mips");

  // python_bindings.cpp:67-86: builder and index names of the unfiltered Vamana variants
  add_vamana<WANN_METRIC_L2, WANN_DTYPE_F32>(m, "float_euclidian", "VamanaFloatEuclidianIndex");
  add_vamana<WANN_METRIC_MIPS, WANN_DTYPE_F32>(m, "float_mips", "VamanaFloatMipsIndex");
  add_vamana<WANN_METRIC_L2, WANN_DTYPE_U8>(m, "uint8_euclidian", "VamanaUInt8EuclidianIndex");
  add_vamana<WANN_METRIC_MIPS, WANN_DTYPE_U8>(m, "uint8_mips", "VamanaUInt8MipsIndex");
  add_vamana<WANN_METRIC_L2, WANN_DTYPE_I8>(m, "int8_euclidian", "VamanaInt8EuclidianIndex");
  add_vamana<WANN_METRIC_MIPS, WANN_DTYPE_I8>(m, "int8_mips", "VamanaInt8MipsIndex");

  // experiments/wrapper.py:245 spells the uint8 classes "Uint8" while the reference registers "UInt8"
  // (python_bindings.cpp:74-79), so its uint8 constructors raise AttributeError there; both spellings resolve here.
  for (const char *cls : {"PrefilterIndex", "RangeFilterTreeIndex", "PostfilterVamanaIndex", "VamanaRangeFilterTreeIndex",
                          "SuperOptimizedPostfilterTreeIndex"})
    for (const char *metric : {"Euclidian", "Mips"})
      m.attr((std::string(cls) + "Uint8" + metric).c_str()) = m.attr((std::string(cls) + "UInt8" + metric).c_str());

  m.def("device_count", [] { return wann_device_count(); });
  m.def("abi_version", [] { return wann_abi_version(); });
  m.def(
      "build_cache_shard",
      [](int kind, int metric, FArray points, FArray labels, int32_t cutoff, double split, double shift,
         const BuildParams &bp, int shard, int nshards, int threads) {
        if (points.ndim() != 2 || labels.ndim() != 1) throw std::runtime_error("bad shapes");
        wann_build_params wb{bp.max_degree, bp.limit, bp.alpha, bp.cache_path.c_str()};
        const float *pp = points.data();
        const float *lp = labels.data();
        int64_t n = points.shape(0), d = points.shape(1);
        int rc;
        {
          py::gil_scoped_release nogil;
          rc = wann_build_cache_shard(kind, metric, WANN_DTYPE_F32, pp, n, d, lp, cutoff, split, shift, &wb, shard,
                                      nshards, threads);
        }
        if (rc) raise_last("build_cache_shard failed");
      },
      "kind"_a, "metric"_a, "points"_a, "filter_values"_a, "cutoff"_a, "split_factor"_a, "shift_factor"_a,
      "build_params"_a, "shard"_a, "nshards"_a, "threads"_a = 0);
  m.def(
      "raw_beam_search",
      [](int metric, FArray points, py::array_t<int32_t, py::array::c_style | py::array::forcecast> graph_rows,
         int64_t subset_start, FArray queries, py::array_t<int64_t, py::array::c_style | py::array::forcecast> query_ids,
         int64_t beam, int64_t limit, int64_t degree_limit, int device) {
        if (points.ndim() != 2 || graph_rows.ndim() != 2 || queries.ndim() != 2) throw std::runtime_error("bad shapes");
        int64_t n = points.shape(0), d = points.shape(1), sn = graph_rows.shape(0), md = graph_rows.shape(1) - 1;
        int64_t nq = queries.shape(0);
        py::array_t<int32_t> ids({(size_t)nq, (size_t)beam});
        py::array_t<float> dists({(size_t)nq, (size_t)beam});
        py::array_t<int32_t> sizes((size_t)nq);
        py::array_t<int64_t> hops((size_t)nq), cmps((size_t)nq);
        int rc = wann_raw_beam_search(metric, points.data(), n, d, graph_rows.data(), md, subset_start, sn,
                                      queries.data(), nq, query_ids.data(), beam, limit, degree_limit,
                                      ids.mutable_data(), dists.mutable_data(), sizes.mutable_data(),
                                      hops.mutable_data(), cmps.mutable_data(), device);
        if (rc) raise_last("raw_beam_search failed");
        return py::make_tuple(ids, dists, sizes, hops, cmps);
      },
      "metric"_a, "points"_a, "graph_rows"_a, "subset_start"_a, "queries"_a, "query_ids"_a, "beam"_a,
      "limit"_a = 10000000, "degree_limit"_a = 10000, "device"_a = 0);
}
