// wann_device.h -- structures shared by the host side (wann_host.cpp) and the gfx950 kernels
// (wann_kernels.hip).  Device index layout in HBM (see DESIGN.md "Data layout"):
//
//   points   float[n][stride]      label-sorted vectors, rows zero padded to 64 B; shared by every
//                                  partition (a partition is a contiguous slice of the sorted order,
//                                  reference: range_filter_tree.h:115-127 subset[i] = start + i)
//   labels   float[n]              sorted labels
//   decoding uint32[n]             sorted index -> original point id (tree_utils.h:85)
//   graph    int32[rows][rs]       adjacency pool: one row per (partition, local node), rs = maxdeg
//                                  rounded up to 16 ints so every row is 64-B aligned; local
//                                  neighbour ids packed at the front, -1 padded (reference in-memory
//                                  layout: (maxdeg+1) ints, slot 0 = degree, graph.h:122-124)
//   parts    PartDesc[]            per partition: first pool row, sorted-order start, size
#pragma once
#include <stdint.h>

namespace wann {

struct PartDesc {
  int64_t row_base;  // first row of this partition in the adjacency pool
  int32_t start;     // sorted index of local node 0
  int32_t n;         // number of nodes
};

struct IndexView {
  const float *points;
  const float *labels;
  const uint32_t *decoding;
  const int32_t *graph;
  const PartDesc *parts;
  // stand-alone PrefilterIndex (prefiltering.h:33-36)
  const float *fv_sorted;
  const int32_t *fi_sorted;
  // window search tree shape (range_filter_tree.h:103): level l has level_nb[l] buckets whose
  // offsets are wst_off[wst_ptr[l] .. wst_ptr[l] + level_nb[l]] (inclusive end)
  const int64_t *wst_off;
  const int64_t *wst_ptr;
  const int64_t *level_part0;  // index of the first partition of level l in parts[]
  const int64_t *level_nb;
  // super tree (super_optimized_postfilter_tree.h:90-91)
  const int64_t *sup_size;
  const int64_t *sup_shift;
  int64_t n;
  int32_t d, stride, rs, maxdeg, metric, kind, nlevels, cutoff, split, vamana_leaves;
  int32_t dtype;  // element type of `points` (wann.h WANN_DTYPE_*): float32 rows, or uint8 / int8 rows of d bytes padded to a
                  // multiple of 64 -- `stride` counts 32-bit words in every case and `points` is addressed in words
};

enum { T_EMPTY = 0, T_GRAPH = 1, T_BRUTE = 2, T_BRUTE_GATHER = 3, T_PARENT = 4 };
// Task::flags: 1 = heavy (schedule first), 2 = final_beam_multiply forced to 1, 4 = speculative sub-task, 8 = mid priority
// (a = beam level, b = parent task index); a T_PARENT task has a = number of sub-tasks, b = first sub-task slot
enum { M_OPTIMIZED = 0, M_THREE_SPLIT = 1, M_FENWICK = 2 };

// One unit of device work: search partition `part` (T_GRAPH) or scan sorted rows [a,b) (T_BRUTE)
// for query `query` with label window [lo,hi].
struct Task {
  int32_t query;
  int32_t mode;
  int32_t part;
  int32_t flags;
  int64_t a, b;
  float lo, hi;
};

struct Counters {
  unsigned long long beam_searches, hops, dist_cmps, brute_rows, label_reads, unsupported;
  // searches run speculatively at beams beyond the one the sequential loop stops at (extra work,
  // not part of the reference's operation count)
  unsigned long long spec_searches, spec_hops, spec_dist_cmps;
  unsigned long long poll_timeouts;  // pollers that gave up waiting (serialised launches); the host re-queues what they left unserved
  // dense prefilter path: queries scored on the matrix cores / of those, sent on to the exact scan / settled by an exact
  // scan of a few 64-position blocks
  unsigned long long gemm_queries, gemm_unproven, gemm_rescued;
  unsigned long long deep_handoffs;  // chains handed to an idle poller of the companion launch (SearchArgs::handoff_beam)
  unsigned long long lookaheads_used;  // levels of a chain that a poller had searched ahead by the time the chain needed them
  // one-wave kernel (wave_beam_search_big): searches, hops, hops served by a helper packet, hops with scoring in the search wave
  unsigned long long big_searches, big_hops, packet_hops, own_scorings, prefetched_hops;
  unsigned long long lookaheads_issued;  // look-ahead searches handed to pollers (used or not)
  unsigned long long empty_windows;      // tree classes: queries whose window lies outside the index's label range (the reference prints a line for each)
};

struct RouteArgs {
  IndexView ix;
  const float *ranges;  // nq x 2
  int64_t nq;
  int32_t method;
  int32_t k;
  int32_t beam, max_beam;
  int32_t has_ratio;
  float ratio;
  Task *tasks;         // [nq * maxt]: query q owns tasks[q*maxt .. q*maxt + qtask_cnt[q])
  int32_t maxt;        // task slots per query (1 unless the method covers a window with several buckets)
  int32_t *qtask_cnt;  // [nq]
  int32_t *graph_list, *graph_count;
  int32_t *heavy_list, *heavy_count;  // graph tasks expected to need several doublings
  // "evidence first": the SMALL speculated levels of tasks that also have levels in the companion launch are served before
  // everything else (what they find tells early whether the task will outgrow its speculated levels); they are stored from
  // the END of heavy_list downwards (heavy_cap entries), prio_count of them
  int32_t *prio_count;
  int32_t heavy_cap;
  int32_t *mid_list, *mid_count;      // graph tasks whose first beam may well fail (expected in-window entries < 4k): served second
  int32_t heavy_ratio;                // partition size / window size at or above which a task is heavy
  int32_t *risk_count;                // graph tasks whose predicted beam (k * partition / window) reaches half of cap_inkernel:
                                      // they may have to double beyond it (the host then starts the continuation pollers)
  int32_t *brute_list, *brute_count;
  // speculative doubling: a heavy task spawns one sub-task per beam level b0 << r, r = 0 .. nsub-1,
  // in the slots [sub_base0, sub_cap); the parent is resolved by whichever sub-task finishes last
  int32_t spec;         // 0 = off
  int32_t spec_num;     // spawn levels up to the first beam with beam * w / n_p >= k * spec_num / 8
  int32_t spec_extra;   // >= 3: a task with at least that many predicted levels whose last level expects fewer than 2 k in-window entries
                        // gets the next level too, if the four-wave kernel runs it (0: never)
  int32_t cap_inkernel;
  int32_t sub_base0, sub_cap;
  int32_t *sub_count;   // next free sub-task slot (relative to sub_base0)
  // levels whose beam exceeds cap_inkernel but not big_cap go to big_list: they are searched by single
  // waves that own a whole workgroup's LDS (k_search "big" workgroups), concurrently with everything else
  int32_t big_cap;      // 0 = no such levels
  int32_t *big_list, *big_count;  // two classes, longest searches first: beams >= 4096 in big_list[0..), the
  int32_t big_stride;             // others in big_list[big_stride..); big_count[2]
  // the speculating tasks with levels in big_list, one entry (the parent's task slot) each: what the idle pollers of
  // k_search scan for chains that will outgrow their levels (may be null)
  int32_t *scan_list, *scan_count;
  Counters *ctr;
  // QueryParams::verbose on the tree classes (range_filter_tree.h:452-457, super_optimized_postfilter_tree.h:226-249): what the
  // descent of a query's window printed, as kVRouteWords words per query -- word 0 = words used, then entries of seven words
  // (kind, index of the query's next task, five arguments): kind 1 "Testing bucket a", kind 2 "Query range = (a,b), smallest
  // containing range (size e) = (c,d)", kind 3 the super tree's descent has ended (its two timing lines), kind 4 the
  // Fenwick search's "Query range: a b" (range_filter_tree.h:363-366), kind 5 its "Searching bucket: a b" (:371-377).  Null otherwise.
  int64_t *vroute;
};
constexpr int kVRouteWords = 1 + 7 * 40;

struct SearchArgs {
  IndexView ix;
  const float *queries;  // (nq, d) row-major, unpadded
  int64_t qid_base;
  Task *tasks;  // (k_search adds look-ahead tasks)
  const int32_t *list;
  const int32_t *list_count;
  const int32_t *heavy_list;   // served before `list` (may be null)
  const int32_t *heavy_count;
  const int32_t *prio_list;    // prio_count entries stored from prio_list[heavy_cap - 1] downwards, served first of all
  const int32_t *prio_count;   // (may be null)
  int32_t heavy_cap;
  const int32_t *mid_list;     // served between the two (may be null): a late doubling would be the tail of the launch
  const int32_t *mid_count;
  int32_t *cursor;
  int32_t B;             // first beam of every task of this launch
  int32_t k;
  int64_t limit;
  int32_t degree_limit;
  int32_t mult;          // final_beam_multiply
  int32_t max_beam;      // postfiltering_max_beam
  int32_t cap_inkernel;  // largest beam this launch may run; beyond it tasks leave via next/final lists
  int32_t pool_bytes;    // per-wave LDS pool for beam + seen-filter
  int32_t is_final;      // this launch is a final re-search: one search, then done
  int32_t *next_list, *next_count;                 // tasks whose next doubling exceeds cap_inkernel
  int32_t *final_list, *final_count;  // tasks whose final re-search beam (recorded in next_beam) exceeds cap_inkernel
  unsigned long long *out_key;  // [ntasks][k]  (fkey(dist) << 32 | sorted id)
  int32_t *out_cnt;             // [ntasks]
  int32_t *g_table;             // per wave slot seen-filter, 4 << g_table_bits bytes each (or null)
  int32_t g_table_bits;
  // wave_beam_search_big (beam in the LDS, filter in g_table): filter entries are tagged with the slot's search epoch
  // (g_epoch[slot], 1..254; the host zeroes table + epochs together), and the exact set of scored nodes is a bitmap of
  // g_seen_words 32-bit words per slot (>= largest partition / 32, a multiple of 4)
  int32_t *g_epoch;
  uint32_t *g_seen;
  int64_t g_seen_words;
  int32_t old_general;          // dev / test: the first-generation general core (wave_beam_search) instead
  int32_t cut_k;                // raw mode, unfiltered VamanaIndex queries: QueryParams::k and ::cut of beamSearch.h:159-167
  double cut;                   //   (0: no cut step -- the post-filter path never takes it)
  int32_t helper;               // one-wave kernel: number of helper waves per workgroup (0 or kHelpers) that prepare row + distance
                                // packets ahead of the search (score_helper)
  unsigned long long *g_beam;   // per wave slot beam, g_beam_cap entries each (or null)
  int64_t g_beam_cap;
  Counters *ctr;
  // raw mode (wann_raw_beam_search): one search at beam B, dump the whole beam
  int32_t raw;
  int32_t *raw_ids;
  float *raw_dists;
  int32_t *raw_sizes;
  long long *raw_hops, *raw_cmps;
  const long long *raw_qids;  // the "own id" of each query (raw mode: Point::id(); wann_batch_search_device_ids); null: qid_base + row
  unsigned long long *prof;   // dev tool: 5 per-phase cycle counters (or null)
  long long *trace;           // dev tool: trace[0] = records, then {task, beam, start, end} per search in 100 MHz ticks (or null)
  int32_t force_general;      // dev / test: never take the small-beam register path
  int32_t *par_done;          // [task slots] finished sub-tasks of a speculating parent
  long long *sub_hops, *sub_cmps;  // [task slots] work of a sub-task (attributed at resolution)
  // companion "big" launch (k_search<METRIC, true>): one wave per workgroup with a large LDS pool serves
  // big_list (speculative levels beyond the ordinary kernel's cap_inkernel, beams up to big_cap)
  int32_t big_cap;           // ordinary kernel: continuations up to this beam may go to a poller (0: none)
  const int32_t *big_list;   // two classes (see RouteArgs)
  const int32_t *big_count;
  int32_t big_stride;
  int32_t *big_cursor;
  // tasks leaving through next_list / final_list record the beam they continue with; a follow-up launch
  // starts each of its tasks at start_beam[task] (null: at B)
  int32_t *next_beam;
  const int32_t *start_beam;
  // Continuations: an ordinary wave whose task must double beyond cap_inkernel (its speculative levels all
  // failed) hands it to a "poller" -- one of the first npollers workgroups of the big launch, which after the
  // static big list waits for such items until every ordinary ticket is done -- instead of to a follow-up launch.
  int32_t npollers;
  int32_t yield_for_big;  // ordinary launch: workgroups [0, #big items) exit at once (room for the companion launch) ...
  int32_t *big_resident;  // ... unless that many companion workgroups are running already (they count themselves here)
  int32_t *dyn_list;    // [tasks], preset to -1; a poller that takes item t leaves -2 - t (the host re-queues entries >= 0)
  int32_t force_poll_timeout;  // test hook: pollers give up at once (exercises the host's recovery of unserved continuations)
  int32_t *dyn_count, *dyn_cursor;
  int32_t *poll_waiting;  // pollers inside their wait loop right now
  int32_t *done_count;  // ordinary tickets completed
  // Deep chains: a task that is about to search at a beam >= handoff_beam (its third doubling level, say) is handed to an
  // IDLE poller if there is one: the poller's search wave has a CU to itself, this wave shares its CU with seven others and
  // would take several times as long over a chain that ends the launch (0 = off)
  int32_t handoff_beam;
  // Look-ahead searches (k_search): task slots for them come from la_count (the sub-task counter of k_route carries on),
  // slots la_base0 + [0, ...) below la_cap; chains at beams >= la_min_beam only.  nullptr = off.
  int32_t *la_count;
  int32_t la_base0, la_cap, la_min_beam;
  int32_t scan_num;  // the scan asks for a look-ahead when the highest level is expected to find at most k * scan_num / 8 entries
  const int32_t *scan_list, *scan_count;  // idle pollers scan these speculating tasks (k_route's list) for ones that will
  int32_t scan_min_top;                   // need the level after their highest one (highest beam >= scan_min_top); null = no scan
  int32_t la_found_max;  // a chain asks for a look-ahead when its last level found fewer in-window entries than this (0.4 k)
  // QueryParams::verbose (postfilter_vamana.h:155-185,230): one record per search of a task's doubling loop -- beam << 42 |
  // unfiltered beam size << 21 | in-window entries of the WHOLE final beam -- in vlog[task * vlog_cap + i], i < vlog_n[task]
  // (one-wave legacy kernel only: the host routes a verbose call there, without speculative levels); null = off
  unsigned long long *vlog;
  int32_t *vlog_n;
  int32_t vlog_cap;
};

struct OrderArgs {  // k_order_heavy: `in` (count entries: task slots) reordered into `out`
  const Task *tasks;
  const int32_t *in;
  int32_t *out;
  const int32_t *count;
};

struct BruteArgs {
  IndexView ix;
  const float *queries;
  const Task *tasks;
  const int32_t *list;
  const int32_t *list_count;
  int32_t *cursor;
  int32_t k;
  unsigned long long *out_key;
  int32_t *out_cnt;
  Counters *ctr;
  // split scans (short lists): partial top-k lists [list position][slice][k rounded to even], their lengths, and one
  // arrival counter per list position (zero between batches); part_key == nullptr: never split
  unsigned long long *part_key;
  int32_t *part_cnt, *part_done;
  int64_t part_cap, part_slots;  // capacity of part_key (keys) and of part_cnt (slices)
};

struct CostArgs {  // k_task_cost: predicted work of every query of a routed batch (hops; an exact scan's rows count 1/36 each)
  const Task *tasks;
  const int32_t *qtask_cnt;
  const PartDesc *parts;
  int64_t nq;
  int32_t maxt, k, beam, max_beam, mult;
  float *cost;  // [nq]
};

struct FinalizeArgs {
  IndexView ix;
  const Task *tasks;
  int32_t maxt;
  const int32_t *qtask_cnt;
  const unsigned long long *out_key;
  const int32_t *out_cnt;
  int64_t nq;
  int32_t k;
  int32_t decode;
  uint32_t pad_id;
  uint32_t *ids;
  float *dists;
};

// launchers implemented in wann_kernels.hip (all asynchronous on `stream`)
struct LaunchCfg {
  int blocks;
  int waves_per_block;  // 0 = kWavesPerBlock
  int big;              // the companion kernel for levels beyond cap_inkernel (k_search<METRIC, true>)
};
int launch_route(const RouteArgs &a, void *stream);
int launch_order_heavy(const OrderArgs &a, void *stream);
// a one-thread kernel that ends when *resident >= need (or after ~0.25 ms): orders the ordinary launch behind the companion's start
int launch_gate(const int32_t *resident, int need, void *stream);
int launch_search(const SearchArgs &a, const LaunchCfg &cfg, void *stream);
// workgroups of that launch the runtime expects to be resident per CU (hipOccupancyMaxActiveBlocksPerMultiprocessor); -1 on error
int search_occupancy(const SearchArgs &a, const LaunchCfg &cfg);
int launch_brute(const BruteArgs &a, int blocks, void *stream);
int launch_finalize(const FinalizeArgs &a, void *stream);
int launch_task_cost(const CostArgs &a, void *stream);
// bytes of LDS one wave of k_search needs: common scratch + the beam / seen-filter pool
int search_lds_bytes_per_wave(int stride, int pool_bytes);
const char *launch_last_error();

constexpr int kWavesPerBlock = 4;
constexpr int kMaxLdsBits = 12;       // build kernels: seen-filters up to 2^12 entries live in the LDS
constexpr int kSearchPoolBytes = 18432;  // k_search per-wave LDS pool: beam + filter of beams <= 128 (2^12 slots), beam alone <= 2304
constexpr int kInKernelBeamCap = 1280;   // largest beam the first (in-kernel doubling) launch runs
constexpr int kHeavyRatio = 8;           // a task whose partition is >= 8x its window is "heavy": served first, its levels speculated
constexpr int kSpecExtraLevels = 2;      // RouteArgs::spec_extra (k_route: one more speculated level for short chains under wide windows)
// one-wave kernel: helper waves per search wave (wann_wave.h score_helper) and the bytes of their LDS mailbox (ScoreBox,
// at the end of the search wave's pool)
constexpr int kHelpers = 3;
constexpr int kScoreBoxBytes = 9600;

}  // namespace wann
