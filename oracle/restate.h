/*
 * oracle/restate.h -- TEST INFRASTRUCTURE ONLY.
 *
 * C ABI of the CPU restatement of JoshEngels/RangeFilteredANN's window-filtered search path.
 * Nothing under rangefilteredann_amd/ (the product) may include, link or call this; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only as the checker.
 *
 * Parity pin: the restatement is checked against the REAL reference compiled from
 * /root/reference (oracle/Makefile `ref` target -> oracle/_ref/) and against golden fixtures that
 * the real reference produced (tests/golden/, generator: tests/golden/make_golden.py).
 * The reference's own tests hold no vectors for this path.
 *
 * All file:line citations are relative to the reference checkout.
 */
#pragma once
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_L2 = 0, ORC_MIPS = 1, ORC_INTEGER = 2 /* | : integer-valued rows of a uint8 / int8 point set, int32 distances */ };

/* index kinds (python_bindings/python_bindings.cpp:111-157) */
enum {
  ORC_PREFILTER = 0,       /* PrefilterIndex<T,Point>                  (src/prefiltering.h:28)   */
  ORC_POSTFILTER = 1,      /* PostfilterVamanaIndex<T,Point>           (src/postfilter_vamana.h) */
  ORC_TREE_PREFILTER = 2,  /* RangeFilterTreeIndex<T,Point>            (prefilter leaves)        */
  ORC_TREE_VAMANA = 3,     /* RangeFilterTreeIndex<..,PostfilterVamana> "VamanaRangeFilterTree"  */
  ORC_SUPER = 4            /* SuperOptimizedPostfilterTree<..,PostfilterVamanaIndex>             */
};

/* QueryParams (ParlayANN/algorithms/utils/types.h:115-140) */
typedef struct {
  int64_t k, beam, limit, degree_limit, final_beam_multiply, max_beam;
  double cut;
  int32_t has_ratio; /* min_query_to_bucket_ratio.has_value() */
  float ratio;
  int32_t verbose;
} orc_qparams;

typedef struct orc_index orc_index;

uint64_t orc_hash64_2(uint64_t x);                 /* parlay/utilities.h:145-150 */
void orc_random_permutation(int64_t n, int32_t *out); /* parlay/random.h:155-159 (default generator) */
int orc_hash_bits(int64_t beam);                   /* beamSearch.h:66 */
float orc_distance(int metric, const float *point, const float *query, uint32_t d);

/* Raw beam search over one graph (beamSearch.h:51-184).  `graph` = n rows of (maxdeg+1) int32,
 * slot 0 = degree (graph.h:122-124,198).  `points` = row-major floats, row `subset_start + local`,
 * `stride` floats per row (zero padded to a multiple of 8).  Returns the beam size; fills the
 * beam (ids/dists) and optionally the visited list (sorted by (dist,id)). */
int64_t orc_beam_search(const int32_t *graph, int64_t n, int64_t maxdeg, const float *points,
                        int64_t stride, int64_t d, int metric, int64_t subset_start,
                        const float *query, int64_t query_id, int64_t start_node, int64_t k,
                        int64_t beam, double cut, int64_t limit, int64_t degree_limit,
                        int32_t *out_ids, float *out_dists, int32_t *vis_ids, float *vis_dists,
                        int64_t *n_visited, int64_t *dist_cmps);

/* Graph cache files (graph.h:126-196): [n:i32][maxDeg:i32][deg[n]:i32][edges:i32...]. */
int orc_graph_load(const char *path, int32_t **rows, int64_t *n, int64_t *maxdeg);
int orc_graph_save(const char *path, const int32_t *rows, int64_t n, int64_t maxdeg);
void orc_free(void *p);

/* Vamana build of ONE partition (vamana/index.h:123-313).  Same batch schedule, beam search,
 * robustPrune, reverse-edge step, final neighbour sort and insertion order
 * (parlay::random_permutation, orc_random_permutation) as the reference.  Exact distance ties
 * break by id (the reference: whatever order libstdc++'s std::sort leaves), so the graph equals
 * the reference builder's whenever no two candidates are exactly equidistant -- pinned by
 * tests/golden/build_golden.npz and tests/test_oracle_vs_reference.py. */
int orc_vamana_build(const float *points, int64_t stride, int64_t d, int metric,
                     int64_t subset_start, int64_t n, int64_t R, int64_t L, double alpha,
                     int32_t *rows /* n*(R+1) */, int threads);

/* Index objects.  points (n,d) f32 row-major, labels (n) f32.  cache_path semantics as
 * postfilter_vamana.h:54-78,126-132: "" = always build; otherwise load
 * cache_path + "vamana_<L>_<R>_<alpha>_<lo>_<hi>_<n>.bin" if present, else build and save. */
orc_index *orc_index_create(int kind, int metric, const float *points, int64_t n, int64_t d,
                            const float *labels, int32_t cutoff, double split_factor,
                            double shift_factor, int64_t R, int64_t L, double alpha,
                            const char *cache_path, int threads);
void orc_index_destroy(orc_index *);
const char *orc_last_error(void);

/* batch_search (range_filter_tree.h:62-96, super_optimized_postfilter_tree.h:60-87,
 * postfilter_vamana.h:191-219, prefiltering.h:124-146).  ranges = nq x 2 (lo,hi) floats.
 * method: "optimized_postfilter" | "three_split" | anything else = fenwick (tree kinds only).
 * counters (optional, 3 x int64): [0] beam searches, [1] hops(visited), [2] distance evals. */
int orc_batch_search(orc_index *, const float *queries, const float *ranges, int64_t nq,
                     const char *method, const orc_qparams *qp, uint32_t *ids, float *dists,
                     int threads, int64_t *counters);

/* Introspection for tests. */
int64_t orc_num_levels(const orc_index *);
int64_t orc_level_size(const orc_index *, int64_t level);
/* partition (level, idx) -> [start,end) in sorted order */
int orc_partition_range(const orc_index *, int64_t level, int64_t idx, int64_t *start, int64_t *end);
const int32_t *orc_partition_graph(const orc_index *, int64_t level, int64_t idx, int64_t *n,
                                   int64_t *maxdeg);
const int64_t *orc_decoding(const orc_index *); /* sorted index -> original id (tree kinds) */
const float *orc_sorted_labels(const orc_index *);

#ifdef __cplusplus
}
#endif
