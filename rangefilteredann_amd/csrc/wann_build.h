// wann_build.h -- host side of index construction: label sort, tree shapes, graph cache files and
// the multi-threaded Vamana builder.  Reference behaviour being reproduced (not its code):
//   label sort / id mapping      src/tree_utils.h:39-98
//   B-ary window search tree     src/range_filter_tree.h:129-189
//   overlapping "super" tree     src/super_optimized_postfilter_tree.h:118-171
//   graph cache name + format    src/postfilter_vamana.h:126-132, ParlayANN/algorithms/utils/graph.h:126-196
//   Vamana batch build           ParlayANN/algorithms/vamana/index.h:61-135,211-313
#pragma once
#include <stdint.h>

#include <functional>
#include <string>
#include <vector>

namespace wann {

struct HostGraph {
  int64_t n = 0;
  int32_t maxdeg = 0;
  std::vector<int32_t> rows;  // n x (maxdeg+1), slot 0 = degree (the reference's in-memory layout)
  int32_t *row(int64_t i) { return rows.data() + i * (maxdeg + 1); }
  const int32_t *row(int64_t i) const { return rows.data() + i * (maxdeg + 1); }
};

struct HostPart {
  int64_t start = 0, n = 0;  // slice [start, start+n) of the sorted order
  float lo = 0, hi = 0;      // min / max label of the slice
  HostGraph g;               // Vamana leaves only
};

struct BuildSpec {
  int kind = 0, metric = 0;
  int dtype = 0;  // element type of the rows (wann.h WANN_DTYPE_*): float32, or uint8 / int8 bytes
  int64_t n = 0, d = 0, stride = 0;
  int32_t cutoff = 1000;
  double split_factor = 2, shift_factor = 0.5;
  int64_t R = 64, L = 500;
  double alpha = 1.0;
  std::string cache;
  int threads = 0;
};

struct HostIndex {
  BuildSpec spec;
  bool vamana_leaves = false, sorted = false;
  std::vector<float> pts;          // n x stride 32-bit words: zero padded rows (label-sorted for tree kinds); float32 values, or the
                                   // d bytes of a uint8 / int8 row
  std::vector<float> labels;       // same order as pts
  std::vector<uint32_t> decoding;  // row -> original id
  std::vector<std::vector<int64_t>> offsets;  // WST bucket offsets per level
  std::vector<int64_t> sup_size, sup_shift;   // super tree per level
  std::vector<std::vector<HostPart>> levels;
  std::vector<float> fv_sorted;    // stand-alone prefilter
  std::vector<int32_t> fi_sorted;
};

// parallel-for over [0,n) on a persistent pool (dynamic chunks); nested calls run inline
void parallel_for(int64_t n, int threads, const std::function<void(int64_t)> &f);
int default_threads();

// metric: bit 0 = inner product, bits 4-5 = element type of the rows behind p / q (0 float32, 1 uint8, 2 int8)
float host_distance(int metric, const float *p, const float *q, int d);

bool graph_file_load(const std::string &path, HostGraph &g);
bool graph_file_save(const std::string &path, const HostGraph &g);
std::string graph_file_name(const BuildSpec &s, float lo, float hi, int64_t n);

// Build the graph of one partition (rows [start, start+n) of pts).
// parlay::random_permutation<int>(n): the reference's insertion order (vamana/index.h:233, parlay/random.h:79-159)
std::vector<int32_t> insertion_order(int64_t n);

void vamana_build(const float *pts, int64_t stride, int64_t d, int metric, int64_t start, int64_t n,
                  int64_t R, int64_t L, double alpha, HostGraph &g, int threads);

// Sort, lay out the tree and obtain every partition's graph (cache or build).
// shard/nshards >= 0: only materialise (build + save) partitions p with p % nshards == shard.
// pending != nullptr: graphs missing from the cache are NOT built here but returned in *pending
// (the caller builds them, on the GPU or with build_pending_on_host, then calls save_built_graphs).
void build_host_index(HostIndex &H, const void *points, const float *labels, int shard = -1,
                      int nshards = 0, std::vector<HostPart *> *pending = nullptr);
void build_pending_on_host(HostIndex &H, std::vector<HostPart *> &pending);
void save_built_graphs(HostIndex &H, std::vector<HostPart *> &built, bool keep);

}  // namespace wann
