/*
 * include/wann.h -- C ABI of the MI355X-native window-filtered ANN engine (libwann.so).
 *
 * This is the drop-in boundary for the ONE hot path of JoshEngels/RangeFilteredANN that this
 * repository accelerates: `batch_search` on the reference's window-search index classes.
 * Plain pointers and sizes only (no torch / pybind11 / STL types); the pybind11 module
 * `window_ann` (rangefilteredann_amd/csrc/window_ann_pybind.cpp) is a thin shim over it and
 * re-exports the reference's Python names.  All file:line citations are into the reference.
 *
 *   wann_index_create        replaces the constructors of
 *       PrefilterIndex<T,Point>                              python_bindings.cpp:111-117, src/prefiltering.h:76-122
 *       PostfilterVamanaIndex<T,Point>                       python_bindings.cpp:129-134, src/postfilter_vamana.h:91-124
 *       RangeFilterTreeIndex<T,Point>                        python_bindings.cpp:119-127, src/range_filter_tree.h:50-59
 *       RangeFilterTreeIndex<T,Point,PostfilterVamanaIndex>  python_bindings.cpp:136-145, src/range_filter_tree.h:50-59,129-189
 *       SuperOptimizedPostfilterTree<..,PostfilterVamanaIndex> python_bindings.cpp:147-157, src/super_optimized_postfilter_tree.h:45-58,118-171
 *   wann_batch_search        replaces <Index>::batch_search
 *       src/range_filter_tree.h:62-96, src/super_optimized_postfilter_tree.h:60-87,
 *       src/postfilter_vamana.h:191-219, src/prefiltering.h:124-146
 *   wann_batch_search_device the same call with queries / ranges / outputs already resident in HBM
 *   wann_batch_search_device_async / wann_wait   that call without blocking: two batches in flight (src/range_filter_tree.h:62-96)
 *   wann_batch_search_device_ids   that call for queries that are not a contiguous range of their batch: per-query own ids
 *       (ParlayANN/algorithms/utils/beamSearch.h:128, src/range_filter_tree.h:71-72; src/postfilter_vamana.h:161-181 for what it is for)
 *   wann_query_params        QueryParams   ParlayANN/algorithms/utils/types.h:115-140, python_bindings.cpp:204-209
 *   wann_build_params        BuildParams   ParlayANN/algorithms/utils/types.h:77-112,  python_bindings.cpp:211-213
 *
 * Error model: functions return 0 on success, non-zero on failure; wann_last_error() gives the
 * message (thread local).  There is NO CPU fallback: every entry point that computes fails with
 * WANN_ERR_NO_DEVICE when no gfx950 device is usable.
 */
#ifndef WANN_H
#define WANN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WANN_ABI_VERSION 5
#define WANN_MAX_DEGREE 128

enum { WANN_OK = 0, WANN_ERR_INVALID = 1, WANN_ERR_NO_DEVICE = 2, WANN_ERR_HIP = 3, WANN_ERR_IO = 4,
       WANN_ERR_UNSUPPORTED = 5 };

/* distance: Euclidian_Point (squared L2, euclidian_point.h:62-75) / Mips_Point (-<q,p>, mips_point.h:60-76) */
enum { WANN_METRIC_L2 = 0, WANN_METRIC_MIPS = 1 };
/* element type of points/queries (python_bindings.cpp:232-237): float32 rows, or uint8 / int8 BYTE rows scored with
 * v_dot4 into exact int32 sums (euclidian_point.h:44-60, mips_point.h:44-58), any dimension */
enum { WANN_DTYPE_F32 = 0, WANN_DTYPE_U8 = 1, WANN_DTYPE_I8 = 2 };
/* index classes */
enum {
  WANN_KIND_PREFILTER = 0,      /* PrefilterIndex                                   */
  WANN_KIND_POSTFILTER = 1,     /* PostfilterVamanaIndex (one graph, raw point ids)  */
  WANN_KIND_TREE_PREFILTER = 2, /* RangeFilterTreeIndex with PrefilterIndex leaves   */
  WANN_KIND_TREE_VAMANA = 3,    /* VamanaRangeFilterTreeIndex (B-ary window search tree) */
  WANN_KIND_SUPER = 4           /* SuperOptimizedPostfilterTreeIndex                 */
};

typedef struct {
  int64_t k;                      /* QueryParams::k                      */
  int64_t beam_width;             /* QueryParams::beamSize               */
  double cut;                     /* QueryParams::cut (inactive on the post-filter path) */
  int64_t limit;                  /* visit limit                         */
  int64_t degree_limit;
  int64_t final_beam_multiply;
  int64_t postfiltering_max_beam;
  int32_t has_min_query_to_bucket_ratio; /* std::optional<float>::has_value() */
  float min_query_to_bucket_ratio;
  int32_t verbose;                /* the doubling loop of every search is dumped to stdout in the reference's words
                                     (postfilter_vamana.h:155-185,230); such a call runs in the one-wave legacy kernel */
} wann_query_params;

typedef struct {
  int64_t max_degree;     /* R; at most 128 (WANN_MAX_DEGREE).  Up to 64 an adjacency row is one 64-lane load and one pass of the seen
                           * filter per hop (every kernel).  64 < R <= 128 (the reference accepts any R: types.h:77-112,
                           * graph.h:115-124,198): rows of up to 128 ids are worked in two halves per hop by the general core of the
                           * one-wave kernel -- same rows as the reference, lower throughput -- and such graphs are built by the host
                           * builder.  A larger R is refused with WANN_ERR_UNSUPPORTED, never truncated.  Every shipped configuration of
                           * the reference's driver uses R = 64 (run_our_method.py:28). */
  int64_t limit;          /* L (build beam) */
  double alpha;
  const char *cache_path; /* graph cache prefix, "" / NULL = none (postfilter_vamana.h:54-78,126-132) */
} wann_build_params;

/* Work counters of the last batch_search (exact, produced by the kernels). */
typedef struct {
  int64_t beam_searches; /* raw beam searches (all doubling rounds + final re-searches) */
  int64_t hops;          /* adjacency rows expanded (= visited nodes)                   */
  int64_t dist_cmps;     /* vectors scored by graph search                              */
  int64_t brute_rows;    /* vectors scored by brute-force scans                         */
  int64_t label_reads;   /* labels read by the post filter                              */
  int64_t rounds;        /* kernel launches of the beam-search kernel                   */
  int64_t spec_searches; /* searches run speculatively beyond the beam the loop stops at */
  int64_t spec_hops;     /*   (extra work of the concurrent doubling levels; not part of  */
  int64_t spec_dist_cmps;/*   the reference's operation count)                            */
  int64_t gemm_queries;  /* PrefilterIndex queries scored through the MFMA GEMM path        */
  double device_ms;      /* HIP-event time of the whole call on its stream              */
  double search_kernel_ms; /* HIP-event time summed over beam-search kernel launches    */
  int64_t recovered_continuations; /* searches a follow-up launch ran because the companion launch's pollers did not
                                      serve them (launches serialised by the runtime / a profiler); 0 normally     */
  int64_t gemm_unproven; /* of gemm_queries: sent on to the exact scan because the MFMA scores could not prove the top k */
  int64_t gemm_rescued;  /* of gemm_queries: proven after an exact scan of a few 64-position blocks of the window        */
  int64_t deep_handoffs; /* search chains that an idle poller of the companion launch (a CU to itself) took over       */
  int64_t lookaheads_used; /* levels of a chain that a poller had searched ahead by the time the chain needed them   */
  /* ABI 3: the one-wave kernel (long searches: a search wave fed by three scoring helper waves) */
  int64_t big_searches;    /* beam searches that ran there (speculated ones included)                                  */
  int64_t big_hops;        /* their hops                                                                               */
  int64_t packet_hops;     /* of big_hops: adjacency row and distances came from a helper wave's packet                */
  int64_t own_scorings;    /* of big_hops: the search wave had to score at least one neighbour itself                  */
  int64_t prefetched_hops; /* of big_hops: packet and filter probes were fetched during the previous hop               */
  int64_t poll_timeouts;   /* pollers of the companion launch that gave up waiting (serialised launches)               */
  int64_t lookaheads_issued; /* look-ahead searches handed to waiting pollers (lookaheads_used of them were needed)    */
} wann_counters;

typedef struct wann_index wann_index;

int wann_abi_version(void);
const char *wann_last_error(void);
/* number of usable gfx950 devices (0 when none); never initialises a context */
int wann_device_count(void);

/* Build (or load from the graph cache) an index and make it resident in the HBM of `device`.
 * points: (n,d) row-major of `dtype` (uint8 / int8 sets stay bytes on the device); labels: (n) float32.  cutoff / split_factor /
 * shift_factor as in the reference constructors (ignored by kinds that have none).
 * build_threads <= 0: PARLAY_NUM_THREADS if set, else all host cores. */
wann_index *wann_index_create(int kind, int metric, int dtype, const void *points, int64_t n,
                              int64_t d, const float *labels, int32_t cutoff, double split_factor,
                              double shift_factor, const wann_build_params *bp, int device,
                              int build_threads);
void wann_index_destroy(wann_index *index);

/* Host-buffer call (the reference's boundary: numpy in, numpy out).  queries: (nq,d) of the index's dtype.  ranges = nq x 2 float32
 * (lo, hi), bounds inclusive (range_filter_tree.h:61).  method: "optimized_postfilter",
 * "three_split", anything else = fenwick (range_filter_tree.h:76-82); ignored by non-tree kinds.
 * ids: nq x k uint32, dists: nq x k float32, caller allocated. */
int wann_batch_search(wann_index *index, const void *queries, const float *ranges, int64_t nq,
                      const char *method, const wann_query_params *qp, uint32_t *ids, float *dists);

/* Device-buffer call: same semantics, every pointer is device memory on the index's device (queries are fp32
 * rows also for uint8 / int8 indexes);
 * `query_id_base` is the global row number of queries[0] (the reference uses the query's row
 * number as its "own id", beamSearch.h:128 + range_filter_tree.h:71-72, so a query shard must
 * keep its global numbering).  Runs on `hip_stream` (a hipStream_t; NULL = the HIP default stream,
 * i.e. ordered after the work the caller queued there) and returns after the stream work is
 * complete.  With a non-blocking stream the caller must make sure that the inputs are complete and
 * that no work still pending on another stream uses the output buffers' memory. */
int wann_batch_search_device(wann_index *index, const void *d_queries, const float *d_ranges,
                             int64_t nq, int64_t query_id_base, const char *method,
                             const wann_query_params *qp, uint32_t *d_ids, float *d_dists,
                             void *hip_stream);

/* wann_batch_search_device for queries that do NOT form a contiguous range of their batch (ABI 5): `d_query_ids[i]` (device array of
 * nq int64) is the global row number of queries[i] -- its "own id" (beamSearch.h:128 + range_filter_tree.h:71-72: the reference never
 * scores the neighbour whose number equals the query's row number).  What a scheduler needs that deals single doubling LEVELS of a
 * query's chain to different GPUs (rangefilteredann_amd/distributed.py level_dealt_batch_search; postfilter_vamana.h:161-172: every
 * level restarts from scratch).  Otherwise as wann_batch_search_device. */
int wann_batch_search_device_ids(wann_index *index, const void *d_queries, const float *d_ranges, int64_t nq,
                                 const int64_t *d_query_ids, const char *method, const wann_query_params *qp,
                                 uint32_t *d_ids, float *d_dists, void *hip_stream);

/* Asynchronous form of wann_batch_search_device (ABI 4).  The reference's call is blocking (src/range_filter_tree.h:62-96: it
 * returns when every query is answered) and so are the two calls above; a serving loop that answers batch after batch leaves
 * the GPU to the tail of one batch and the ramp-up of the next about a fifth of the time at 10 000 queries per batch.  This call
 * returns at once with a ticket: the batch runs on one of the LANES of the index (two; the environment variable WANN_ASYNC_LANES,
 * 1 .. 4, read when the first asynchronous call creates them, sets another depth; a lane = its own per-batch workspace, streams
 * and host worker thread), ordered after the work the caller has queued on `after_stream` so far (its inputs; NULL = the HIP
 * default stream), and concurrently with the other lane's batch.  Same rows as the blocking call.  The output buffers belong
 * to the call until wann_wait(ticket) has returned; wait for ticket t before submitting ticket t + <lanes> (its lane is reused).
 * wann_wait returns the batch's status (message in wann_last_error) and, if `counters` is not NULL, its work counters. */
int wann_batch_search_device_async(wann_index *index, const void *d_queries, const float *d_ranges, int64_t nq,
                                   int64_t query_id_base, const char *method, const wann_query_params *qp, uint32_t *d_ids,
                                   float *d_dists, void *after_stream, int64_t *ticket);
int wann_wait(wann_index *index, int64_t ticket, wann_counters *counters);

/* Predicted work of every query of a batch, in beam-search hops (cost[nq], host; ranges: nq x 2 float32, host): the batch is
 * routed like wann_batch_search routes it (src/range_filter_tree.h:403-471) and every query's tasks are priced with the doubling
 * loop of src/postfilter_vamana.h:161-181 under "a beam finds k entries once it is expected to hold k in-window points".  What a
 * strong-scaling shard cut balances: contiguous shards of equal predicted work instead of equal query counts. */
int wann_predict_costs(wann_index *index, const float *ranges, int64_t nq, const char *method, const wann_query_params *qp,
                       float *cost);

int wann_get_counters(const wann_index *index, wann_counters *out);

/* Introspection (tests, tools). */
int64_t wann_num_points(const wann_index *index);
int64_t wann_dim(const wann_index *index);
int64_t wann_num_levels(const wann_index *index);
int64_t wann_level_size(const wann_index *index, int64_t level);
int wann_partition_range(const wann_index *index, int64_t level, int64_t idx, int64_t *start, int64_t *end);
/* copy partition graph out in the reference's in-memory layout: n x (R+1) int32, slot 0 = degree.  `rows` holds
 * cap_rows x (max_degree+1) ints; max_degree must equal the index's R (wann_max_degree), cap_rows >= partition size. */
int wann_partition_graph(const wann_index *index, int64_t level, int64_t idx, int32_t *rows, int64_t cap_rows,
                         int64_t max_degree);
int64_t wann_max_degree(const wann_index *index);
int64_t wann_device_bytes(const wann_index *index);
/* In-process multi-device mode: with WANN_DEVICES=a,b,... in the environment wann_index_create makes the index resident on
 * every listed device (a device may be listed twice) and wann_batch_search -- the host-buffer call, the reference's boundary
 * (src/range_filter_tree.h:62-96: one call parallelises over all queries) -- cuts its batch into contiguous shards, one host
 * thread + stream per replica; queries keep their global row numbers.  wann_batch_search_device serves the primary only.
 * Returns the number of replicas (1 without WANN_DEVICES). */
int wann_num_replicas(const wann_index *index);

/* The multi-device call with DEVICE-RESIDENT gathered rows (ABI 4), for a host that keeps working on the GPUs: every replica of
 * WANN_DEVICES (distinct devices: one RCCL rank each) searches its contiguous shard -- wann_gather_layout gives shard s's first
 * row, row count and the common capacity cap = ceil(nq / replicas) -- and ONE ncclAllGather over RCCL / xGMI leaves, on EVERY
 * replica's device, the planes of all shards:   d_planes[r] -> int32 [replicas][2][cap][k]   (device memory of replica r, owned
 * by the index, valid until its next call): plane [s][0] = ids (uint32) of shard s's rows, plane [s][1] = their distances
 * (float32 bits); rows beyond a shard's count are the reference's padding (id 0, FLT_MAX).  Global query q of shard s sits at
 * row q - lo_s.  queries / ranges: host buffers like wann_batch_search.  librccl.so is opened at the first call (dlopen). */
int wann_gather_layout(int64_t nq, int world, int shard, int64_t *lo, int64_t *count, int64_t *cap);
int wann_batch_search_allgather(wann_index *index, const void *queries, const float *ranges, int64_t nq, const char *method,
                                const wann_query_params *qp, int32_t **d_planes, int64_t *cap);

/* Graph-cache tool: build (host, multi-threaded) and save only the cache files of the
 * partitions p with p % nshards == shard, without creating a device index.  Used to split the
 * build of one index over the ranks of a multi-GPU job that share a cache directory. */
int wann_build_cache_shard(int kind, int metric, int dtype, const void *points, int64_t n, int64_t d,
                           const float *labels, int32_t cutoff, double split_factor,
                           double shift_factor, const wann_build_params *bp, int shard, int nshards,
                           int build_threads);

/* Raw kernels, exported for parity tests and micro-benchmarks (device pointers). */
/* one beam search per query over ONE graph given in the reference's layout (host pointers;
 * the call uploads, runs the production kernel and downloads). */
int wann_raw_beam_search(int metric, const float *points, int64_t n, int64_t d,
                         const int32_t *graph_rows /* n x (maxdeg+1) */, int64_t maxdeg,
                         int64_t subset_start, int64_t subset_n, const float *queries, int64_t nq,
                         const int64_t *query_ids, int64_t beam, int64_t limit,
                         int64_t degree_limit, int32_t *out_ids /* nq x beam */,
                         float *out_dists /* nq x beam */, int32_t *out_sizes /* nq */,
                         int64_t *out_hops /* nq */, int64_t *out_dist_cmps /* nq */, int device);

/* Unfiltered VamanaIndex<T,Point> (ParlayANN/python/vamana_index.cpp:42-76, bound at python_bindings.cpp:92-109): a graph
 * index opened from a point file (uint32 n, uint32 d, n*d elements of dtype: point_range.h:63-93) and a graph file
 * (graph.h:126-196: the format of the graph cache).  batch_search = beam_search from node 0 with
 * QueryParams(knn, beam_width, 1.35, n, max_degree) -- including the k / cut step of beamSearch.h:159-167, which only this
 * path takes (it lives in the first-generation core of the one-wave legacy kernel, k_search<., 2>: same rows as the reference, a
 * fraction of the filtered path's throughput) -- and the first knn entries of the final beam (a shorter beam is padded with id
 * 2^32-1, FLT_MAX; the reference reads past it).  queries: host (nq,d) of the index's dtype. */
typedef struct wann_vamana wann_vamana;
wann_vamana *wann_vamana_open(int metric, int dtype, const char *data_path, const char *graph_path, int device);
void wann_vamana_close(wann_vamana *index);
int64_t wann_vamana_num_points(const wann_vamana *index);
int64_t wann_vamana_dim(const wann_vamana *index);
int wann_vamana_batch_search(wann_vamana *index, const void *queries, int64_t nq, int64_t knn, int64_t beam_width,
                             uint32_t *ids, float *dists);
/* build_vamana_index (ParlayANN/python/builder.cpp:33-59, bound at python_bindings.cpp:93-95): point file -> graph file,
 * built on the GPU with the reference's insertion order */
int wann_vamana_build_file(int metric, int dtype, const char *data_path, const char *graph_out_path, int64_t max_degree,
                           int64_t limit, double alpha, int device);

#ifdef __cplusplus
}
#endif
#endif /* WANN_H */
