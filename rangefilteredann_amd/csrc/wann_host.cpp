// wann_host.cpp -- host side of the MI355X window-filtered ANN engine: device residency of the index, launch geometry and
// the batch_search driver (routing -> brute scans -> doubling rounds of the beam-search kernel -> final re-search ->
// finalize), the dense prefilter path's launches, the GPU build of missing graphs.  The C ABI (include/wann.h) that calls
// into this is wann_abi.cpp; the raw-graph entry points are wann_raw.cpp.
//
// There is no CPU search path in this library: every compute entry point needs a gfx950 device
// and fails loudly otherwise.
#include "wann_host_internal.h"

namespace wann_host {


thread_local std::string g_err;
int fail(int code, const std::string &msg) {
  g_err = msg;
  return code;
}

int usable_devices() {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int hash_bits(int64_t beam) {  // beamSearch.h:66
  return std::max<int>(10, (int)std::ceil(std::log2((double)(beam * beam))) - 2);
}

// reference in-memory rows -> device rows: rs ints per row, neighbours packed, -1 padded
void convert_rows(const HostGraph &g, int rs, int32_t *out) {
  for (int64_t i = 0; i < g.n; i++) {
    const int32_t *r = g.row(i);
    int32_t *o = out + i * rs;
    const int deg = r[0];
    if (deg < 0 || deg > g.maxdeg || deg > rs) throw std::runtime_error("graph row degree out of range");
    for (int j = 0; j < deg; j++) {
      if (r[1 + j] < 0 || r[1 + j] >= g.n) throw std::runtime_error("graph edge out of range");
      o[j] = r[1 + j];  // a row may list a node twice (reference-built graphs do); the kernel copes
    }
    for (int j = deg; j < rs; j++) o[j] = -1;
  }
}


// the index's switches for one call (WANN_TEST_HOOKS=1: re-read from the environment first)
Tuning snapshot_tuning(wann_index &I) {
  std::lock_guard<std::mutex> lk(I.tune_mu);
  if (I.tune.hooks_live) I.tune = Tuning::from_env();
  return I.tune;
}

void upload_index(wann_index &I) {
  HostIndex &H = I.host();
  const BuildSpec &s = H.spec;
  HIP_CHECK(hipSetDevice(I.device));
  hipDeviceProp_t prop;
  HIP_CHECK(hipGetDeviceProperties(&prop, I.device));
  I.num_cus = prop.multiProcessorCount;
  std::string arch = prop.gcnArchName;
  if (arch.rfind("gfx950", 0) != 0)
    throw HipError("device " + std::to_string(I.device) + " is " + arch + ", this library holds gfx950 code only");
  HIP_CHECK(hipStreamCreateWithFlags(&I.own_stream, hipStreamNonBlocking));
  {  // highest priority: when both launches become ready the companion's few workgroups are dispatched first
    int prio_low = 0, prio_high = 0;
    HIP_CHECK(hipDeviceGetStreamPriorityRange(&prio_low, &prio_high));
    HIP_CHECK(hipStreamCreateWithPriority(&I.side_stream, hipStreamNonBlocking, prio_high));
  }

  IndexView &v = I.view;
  v.n = s.n;
  v.d = (int32_t)s.d;
  v.stride = (int32_t)s.stride;
  v.metric = s.metric;
  v.dtype = s.dtype;
  v.kind = s.kind;
  v.cutoff = s.cutoff;
  v.split = (int32_t)s.split_factor;
  v.vamana_leaves = H.vamana_leaves ? 1 : 0;
  v.maxdeg = (int32_t)s.R;
  v.rs = (int32_t)(((s.R + 15) / 16) * 16);
  v.nlevels = (int32_t)H.levels.size();

  I.d_points.upload(H.pts);
  I.d_labels.upload(H.labels);
  I.d_decoding.upload(H.decoding);
  v.points = I.d_points.p;
  v.labels = I.d_labels.p;
  v.decoding = I.d_decoding.p;
  if (s.kind == WANN_KIND_PREFILTER) {
    I.d_fv.upload(H.fv_sorted);
    I.d_fi.upload(H.fi_sorted);
    v.fv_sorted = I.d_fv.p;
    v.fi_sorted = I.d_fi.p;
  }
  // partitions + adjacency pool
  int64_t rows = 0;
  std::vector<int64_t> level_nb;
  for (auto &lv : H.levels) {
    I.level_part0.push_back((int64_t)I.parts.size());
    level_nb.push_back((int64_t)lv.size());
    for (auto &P : lv) {
      PartDesc pd;
      pd.row_base = rows;
      pd.start = (int32_t)P.start;
      pd.n = (int32_t)P.n;
      I.parts.push_back(pd);
      if (H.vamana_leaves) rows += P.n;
    }
  }
  I.d_parts.upload(I.parts);
  v.parts = I.d_parts.p;
  I.d_level_part0.upload(I.level_part0);
  I.d_level_nb.upload(level_nb);
  v.level_part0 = I.d_level_part0.p;
  v.level_nb = I.d_level_nb.p;
  if (H.vamana_leaves) {
    I.d_graph.ensure((size_t)rows * v.rs);
    // convert + upload partition by partition through a bounded pinned staging buffer
    const size_t stage_rows = 1 << 20;
    std::vector<int32_t> stage;
    size_t pi = 0;
    for (auto &lv : H.levels)
      for (auto &P : lv) {
        const PartDesc &pd = I.parts[pi++];
        if (P.g.n != P.n) {  // not in the cache: built on the device afterwards, rows start empty
          HIP_CHECK(hipMemset(I.d_graph.p + pd.row_base * v.rs, 0xFF, (size_t)pd.n * v.rs * 4));
          continue;
        }
        for (int64_t r0 = 0; r0 < P.n; r0 += (int64_t)stage_rows) {
          int64_t cnt = std::min<int64_t>((int64_t)stage_rows, P.n - r0);
          stage.resize((size_t)cnt * v.rs);
          for (int64_t i = 0; i < cnt; i++) {
            const int32_t *r = P.g.row(r0 + i);
            int32_t *o = stage.data() + i * v.rs;
            const int deg = r[0];
            if (deg < 0 || deg > P.g.maxdeg || deg > v.rs) throw std::runtime_error("graph row degree out of range");
            for (int j = 0; j < deg; j++) {
              if (r[1 + j] < 0 || r[1 + j] >= P.n) throw std::runtime_error("graph edge out of range");
              o[j] = r[1 + j];
            }
            for (int j = deg; j < v.rs; j++) o[j] = -1;
          }
          HIP_CHECK(hipMemcpy(I.d_graph.p + (pd.row_base + r0) * v.rs, stage.data(), stage.size() * 4, hipMemcpyHostToDevice));
        }
      }
    v.graph = I.d_graph.p;
  }
  if (!H.offsets.empty()) {
    std::vector<int64_t> flat, ptr;
    for (auto &o : H.offsets) {
      ptr.push_back((int64_t)flat.size());
      flat.insert(flat.end(), o.begin(), o.end());
    }
    ptr.push_back((int64_t)flat.size());
    I.d_wst_off.upload(flat);
    I.d_wst_ptr.upload(ptr);
    v.wst_off = I.d_wst_off.p;
    v.wst_ptr = I.d_wst_ptr.p;
  }
  if (!H.sup_size.empty()) {
    I.d_sup_size.upload(H.sup_size);
    I.d_sup_shift.upload(H.sup_shift);
    v.sup_size = I.d_sup_size.p;
    v.sup_shift = I.d_sup_shift.p;
  }
  I.device_bytes = (int64_t)(I.d_points.bytes() + I.d_labels.bytes() + I.d_decoding.bytes() + I.d_graph.bytes() +
                             I.d_parts.bytes() + I.d_fv.bytes() + I.d_fi.bytes() + I.d_wst_off.bytes());
}

// Global scratch of a launch whose searches keep their seen-filter in global memory: the per-slot filter tables
// (entries tagged with the slot's search epoch), the epochs, and the per-slot exact seen bitmaps.  A table whose
// slot layout changes (or that was reallocated) is zeroed together with its epochs.
int ensure_filter_scratch(DevBuf<int32_t> &table, DevBuf<int32_t> &epoch, DevBuf<uint32_t> &seen, int64_t &layout, int slots,
                          int table_bits, int64_t seen_words, hipStream_t st, bool any_slots) {
  // any_slots: a slot's region depends on the region stride only (slot << stride bits), so launches with different slot counts
  // share one zeroed table -- and the stride never SHRINKS: a launch whose filters are smaller than the largest seen so
  // far uses a prefix of every region (the search masks with its own `bits`), instead of a 2^bits-word re-layout that zeroes the
  // whole table and every epoch each time two launch shapes alternate.  The stride in use is returned (SearchArgs::g_table_bits).
  if (any_slots && layout > table_bits && layout < 32 && ((size_t)slots << layout) * sizeof(int32_t) <= ((size_t)20 << 30)) table_bits = (int)layout;
  const size_t need = (size_t)slots << table_bits;
  const int64_t want = any_slots ? (int64_t)table_bits : (((int64_t)slots << 8) | table_bits);
  const bool fresh = need > table.cap || (size_t)slots > epoch.cap;
  table.ensure(need);
  epoch.ensure((size_t)slots);
  seen.ensure((size_t)slots * (size_t)seen_words);
  if (fresh || layout != want) {
    HIP_CHECK(hipMemsetAsync(table.p, 0, table.cap * sizeof(int32_t), st));
    HIP_CHECK(hipMemsetAsync(epoch.p, 0, epoch.cap * sizeof(int32_t), st));
    layout = want;
  }
  return table_bits;
}

// Launch geometry for a k_search launch whose searches run beams in [first_beam, cap].
// big_lds: the rare follow-up launch for beams beyond the in-kernel cap -- one wave per workgroup with a
// pool large enough to keep even a 10 000-entry beam in the LDS (only its seen-filter lives in global memory).
// The four-wave kernel holds the register-resident cores and the second-generation general core only (256 registers per
// wave); everything else -- beams whose LDS beam exceeds the four-wave pool, the first-generation cores behind the test
// switches -- runs in the one-wave-per-workgroup kernel (big_lds).  force_table: a global seen-filter even if the LDS would do.
// legacy: the one-wave kernel that holds the first-generation cores (k_search<., 2>: dev switches, the cut step); it is also
// what a beam that does not fit the LDS next to the helper waves' mailbox gets.  The production one-wave kernel
// (k_search<., 1>) always keeps its seen-filter in global memory.
// base_pool: the four-wave kernel's per-wave pool (kSearchPoolBytes, or lean_pool_bytes() for three workgroups per CU).
// A workgroup's LDS is handed out in granules: 1 280 bytes fits what round 4 measured (three workgroups of 53 KiB = 43 granules each
// are not co-resident in a CU's 128 granules, three of 52.5 KiB = 42 are; two of 77.5 KiB = 62 are) -- whatever
// hipOccupancyMaxActiveBlocksPerMultiprocessor says.
constexpr int kLdsGranule = 1280, kLdsPerCu = 160 * 1024;
inline int lds_blocks_per_cu(int per_block) { return kLdsPerCu / (((per_block + kLdsGranule - 1) / kLdsGranule) * kLdsGranule); }

RoundCfg config_for(const wann_index &I, const Tuning &T, int64_t first_beam, int64_t cap, int64_t work_items, bool big_lds, bool force_table, bool legacy,
                    int base_pool) {
  RoundCfg rc{};
  const int64_t cap_bytes = ((cap + 1) & ~(int64_t)1) * 8;
  if (cap_bytes > base_pool) big_lds = true;
  // Rows of more than 64 neighbours (64 < R <= 128) are worked in two halves by the first-generation general core only: every
  // search of such an index runs in the one-wave kernel that holds it (k_search<., 2>), eight workgroups per CU.
  if (I.view.rs > 64) big_lds = legacy = true;
  const int wpb = big_lds ? 1 : kWavesPerBlock;
  const int common = search_lds_bytes_per_wave(I.view.stride, 0);
  if (big_lds && !legacy && cap_bytes + 4096 + kScoreBoxBytes > 150 * 1024 - common) legacy = true;
  const int box_bytes = (big_lds && !legacy) ? kScoreBoxBytes : 0;
  if (big_lds && !legacy) force_table = true;
  int pool = base_pool;
  // (one-wave kernels: + the clash-detection scratch beside the largest beam; production kernel: + the helper waves' mailbox
  // at the end of the pool -- the kernel takes kScoreBoxBytes off whenever it runs with helpers, so the decisions below use
  // what is left)
  if (big_lds) pool = (int)std::min<int64_t>(std::max<int64_t>(cap_bytes + 4096, kSearchPoolBytes) + box_bytes, 150 * 1024 - common);
  rc.pool_bytes = pool;
  const int usable = pool - box_bytes;
  const int per_block = (common + pool) * wpb;
  if (per_block > 160 * 1024) throw std::runtime_error("beam-search LDS footprint exceeds 160 KiB");
  // register budget: the L2 kernel holds two whole 512-B rows per lane pair in flight (2 waves/SIMD)
  // (four-wave kernel: the squared-L2 float kernel needs 232 registers: two waves per SIMD; the inner-product and byte-row
  // kernels are built for three)
  const int waves_per_cu = (I.view.metric == 1 || I.view.dtype != WANN_DTYPE_F32) ? 12 : 8;
  int blocks_per_cu = std::min((big_lds ? 8 : waves_per_cu) / wpb, lds_blocks_per_cu(per_block));
  blocks_per_cu = std::max(1, blocks_per_cu);
  int64_t blocks = (int64_t)I.num_cus * blocks_per_cu;
  const int cap_bits = hash_bits(cap);
  if (force_table || cap_bytes + ((int64_t)4 << cap_bits) > usable) {  // some beam of the range keeps its filter in global memory
    rc.table_bits = cap_bits;
    int64_t per_slot = (int64_t)4 << cap_bits;
    int64_t max_slots = std::max<int64_t>(wpb, ((int64_t)16 << 30) / per_slot);
    blocks = std::min(blocks, max_slots / wpb);
  }
  if (cap_bytes > usable) rc.beam_cap = (cap + 1) & ~(int64_t)1;
  (void)first_beam;
  blocks = std::min<int64_t>(blocks, (work_items + wpb - 1) / wpb);
  rc.lc.blocks = (int)std::max<int64_t>(blocks, 1);
  rc.lc.waves_per_block = wpb;
  rc.lc.big = big_lds ? (legacy ? 2 : 1) : 0;
  rc.big_lds = big_lds;
  rc.slots = rc.lc.blocks * wpb;
  return rc;
}

// Per-wave pool with which THREE four-wave workgroups share a CU's 160 KiB of LDS (inner-product / byte-row kernels, which
// fit three waves per SIMD): the in-kernel cap's beam (10 KiB) still fits, the seen-filter of beams up to 90 too.  0: this
// index can not use it (rows too long).
int lean_pool_bytes(const wann_index &I, const Tuning &T) {
  if (!(I.view.metric == 1 || I.view.dtype != WANN_DTYPE_F32)) return 0;
  const int common = search_lds_bytes_per_wave(I.view.stride, 0);
  // 52 KiB per workgroup.  Three workgroups of 53 KiB (159 of the CU's 160 KiB) are NOT co-resident on gfx950, whatever
  // hipOccupancyMaxActiveBlocksPerMultiprocessor says (3): a launch of 768 such workgroups ran at the speed of 512 until round 4
  // measured it (stand-alone inner-product graph, 30 000 searches at beam 80: 6.65 ms at 53 KiB, 5.22 ms at 52.5 KiB and below).
  int pool = (52 * 1024) / kWavesPerBlock - common;
  return pool >= kInKernelBeamCap * 8 + 1536 ? pool : 0;
}

int method_code(const char *m) {
  if (m && !strcmp(m, "optimized_postfilter")) return M_OPTIMIZED;
  if (m && !strcmp(m, "three_split")) return M_THREE_SPLIT;
  return M_FENWICK;  // range_filter_tree.h:76-82: everything else falls through to fenwick
}

// PrefilterIndex batches in which many queries share a window: those windows are scored as Q x P^T GEMMs on the
// matrix cores (wann_gemm_kernels.hip), ~32 candidates per query are kept and re-ranked exactly; everything else (and
// every query whose top-k cannot be proven from the MFMA scores) goes through the exact scan kernel.  Grouping,
// tile planning and the hand-over to the exact scan all happen on the device: the host enqueues six launches and
// never waits.
void dense_prefilter(wann_index &I, const Tuning &T, Workspace &W, const float *d_queries, int64_t nq, int k, hipStream_t st) {
  if (k > kSelect / 2 || I.view.stride > 512 || (I.view.stride & 15)) return;  // (rows of up to 512 floats: RedCaps)
  if (!I.have_norms) {
    I.d_pnorm2.ensure((size_t)I.view.n);
    I.d_pnorm2_max.ensure(1);
    HIP_CHECK(hipMemsetAsync(I.d_pnorm2_max.p, 0, sizeof(unsigned int), st));
    if (launch_point_norms(I.view, I.d_pnorm2.p, I.d_pnorm2_max.p, st)) throw HipError(std::string("k_point_norms: ") + gemm_launch_last_error());
    I.have_norms = true;
  }
  size_t cap = 64;
  while (cap < (size_t)nq * 2) cap <<= 1;
  I.g_slot_key.ensure(cap);
  I.g_slot_count.ensure(cap);
  I.g_slot_group.ensure(cap);
  I.g_slot_list.ensure((size_t)nq);
  I.g_q_slot.ensure((size_t)nq);
  I.g_q_rank.ensure((size_t)nq);
  I.g_plan.ensure(P_INTS);
  I.g_score_used.ensure(1);
  I.g_groups.ensure((size_t)nq / kGroupMinQueries + 1);
  I.g_gq.ensure((size_t)nq);
  I.g_tq_group.ensure((size_t)nq);
  I.g_tq_local.ensure((size_t)nq);
  // the blocks' hand-over (two blocks of four floats per query and 128 window positions), capped at 256 MiB (groups
  // beyond that take the exact scan)
  // (a window group uses queries x its own blocks x 8 floats: 25 MB for the adversarial batch; what does not fit the cap takes the exact scan)
  const size_t score_cap = (size_t)std::min<unsigned long long>((unsigned long long)nq * (unsigned long long)((I.view.n + 127) / 128) * 8ull, 64ull << 20);
  I.g_scores.ensure(score_cap);
  I.g_tile_group.ensure(score_cap / 1024 + 1);
  GemmArgs ga{};
  ga.ix = I.view;
  ga.queries = d_queries;
  ga.tasks = W.tasks.p;
  ga.nq = nq;
  ga.slot_key = I.g_slot_key.p;
  ga.slot_count = I.g_slot_count.p;
  ga.slot_group = I.g_slot_group.p;
  ga.slot_list = I.g_slot_list.p;
  ga.cap_mask = (int32_t)(cap - 1);
  ga.q_slot = I.g_q_slot.p;
  ga.q_rank = I.g_q_rank.p;
  ga.plan = I.g_plan.p;
  ga.score_used = I.g_score_used.p;
  ga.groups = I.g_groups.p;
  ga.tile_group = I.g_tile_group.p;
  ga.gq = I.g_gq.p;
  ga.tq_group = I.g_tq_group.p;
  ga.tq_local = I.g_tq_local.p;
  ga.pnorm2 = I.d_pnorm2.p;
  ga.pnorm2_max_bits = I.d_pnorm2_max.p;
  ga.scores = I.g_scores.p;
  ga.score_cap = (int64_t)score_cap;
  ga.k = k;
  // fp32 accumulation of the 3 d exact bf16 x bf16 products of a score: worst case (3 d) u |q||p| for ANY order of the
  // additions, u = 2^-24 with a rounding adder, 2^-23 with a truncating one; 3 = the truncating bound and half as much again.
  // (Round 2 used 8: at d = 512 that one term was 7.4e-4 |q||p|, 46 % of the adversarial queries could not be proven.)
  // The knob can only widen the margin (Tuning clamps it to >= 3): a smaller factor would certify unproven results.
  ga.acc_factor = T.proof_factor;
  ga.out_key = W.out_key.p;
  ga.out_cnt = W.out_cnt.p;
  ga.brute_list = W.list_brute.p;
  ga.brute_count = W.ints.p + I_BRUTE_COUNT;
#ifdef WANN_GEMM_PROF
  I.g_prof.ensure(8);
  HIP_CHECK(hipMemsetAsync(I.g_prof.p, 0, 64, st));
  ga.prof = I.g_prof.p;
#endif
  if (launch_group_windows(ga, W.ctr.p, st)) throw HipError(std::string("k_group_*: ") + gemm_launch_last_error());
  if (launch_gemm_scores(ga, I.num_cus, st)) throw HipError(std::string("k_gemm_scores: ") + gemm_launch_last_error());
  if (launch_select_rerank(ga, W.ctr.p, st)) throw HipError(std::string("k_rerank: ") + gemm_launch_last_error());
#ifdef WANN_GEMM_PROF
  unsigned long long h[8];
  HIP_CHECK(hipMemcpyAsync(h, I.g_prof.p, 64, hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  fprintf(stderr, "k_gemm_scores cycles summed over waves: stage %llu barrier %llu fetch+mfma %llu store %llu barrier %llu kernel %llu\n", h[0], h[1], h[2], h[3], h[4], h[5]);
#endif
}

// W / side / last: the lane of this batch (the index's own members for the blocking calls, an AsyncLane's for the asynchronous one)
void run_batch(wann_index &I, Workspace &W, hipStream_t side, wann_counters &last, const float *d_queries, const float *d_ranges, int64_t nq,
               int64_t qid_base, const char *method, const wann_query_params &qp, uint32_t *d_ids, float *d_dists, hipStream_t st, const Tuning &T,
               const int64_t *d_qids) {
  if (qp.k <= 0 || qp.k > 1024) throw std::runtime_error("k must be in [1, 1024]");
  // the brute-force classes ignore the beam (the reference driver passes beam_size = 0 there, run_our_method.py:256)
  if (qp.beam_width <= 0 && I.host().vamana_leaves) throw std::runtime_error("beam_width must be positive");
  if (qp.postfiltering_max_beam > (1 << 20)) throw std::runtime_error("postfiltering_max_beam too large");
  HIP_CHECK(hipSetDevice(I.device));
  std::unique_lock<std::mutex> dense_lock(I.dense_mu, std::defer_lock);
  if (I.host().spec.kind == WANN_KIND_PREFILTER) dense_lock.lock();
  const int k = (int)qp.k;
  const int mcode = method_code(method);
  const bool tree = I.host().spec.kind == WANN_KIND_TREE_PREFILTER || I.host().spec.kind == WANN_KIND_TREE_VAMANA;
  // fenwick / three_split cover a window with several buckets (+ two brute-forced ends)
  // (optimized_postfilter needs one slot unless its tiny-window / ratio fallback reaches the
  // multi-bucket fenwick cover, which cannot happen for split <= 4 without a ratio: SURVEY.md A.5)
  const bool single = !tree || (mcode == M_OPTIMIZED && !qp.has_min_query_to_bucket_ratio && I.host().spec.split_factor <= 4);
  const int maxt = single ? 1 : 96;
  // (T: this call's copy of the index's switches -- snapshot_tuning)
  // QueryParams::verbose (postfilter_vamana.h:155-185,230): the doubling loop of every (query, partition) search is dumped to
  // stdout in the reference's words after the batch -- a debugging aid: such a call runs plain sequential doubling in the
  // one-wave legacy kernel, which records every search
  const bool verbose_call = qp.verbose != 0 && I.host().vamana_leaves;
  // (wide rows, R > 64: plain in-kernel doubling in the one-wave kernel, no speculative levels / companion launch)
  const bool spec = I.host().vamana_leaves && T.spec && I.view.rs <= 64 && !verbose_call;
  const int64_t sub_slots = spec ? std::min<int64_t>(nq * (int64_t)maxt * 4 + 1024, (int64_t)1 << 26) : 0;
  W.ensure(nq, k, maxt, sub_slots);
  if (spec) HIP_CHECK(hipMemsetAsync(W.par_done.p, 0, ((size_t)nq * maxt) * sizeof(int32_t), st));
  // (look-ahead slot of a task: none.  A resolved parent is never reset by a wave before the one-wave kernel reads it.)
  if (spec) HIP_CHECK(hipMemsetAsync(W.sub_cmps.p, 0xFF, ((size_t)nq * maxt) * sizeof(long long), st));
  // (lowest level of a speculating task that has found k entries so far: none -- higher levels become moot, k_search)
  if (spec) HIP_CHECK(hipMemsetAsync(W.sub_hops.p, 0x7F, ((size_t)nq * maxt) * sizeof(long long), st));
  // (a sub-task slot's count is -1 until its search has finished: what the pollers' scan goes by)
  if (spec) HIP_CHECK(hipMemsetAsync(W.out_cnt.p + (size_t)nq * maxt, 0xFF, (size_t)sub_slots * sizeof(int32_t), st));
  last = wann_counters{};
  if (nq == 0) return;
  HIP_CHECK(hipMemsetAsync(W.ints.p, 0, kInts * sizeof(int32_t), st));
  HIP_CHECK(hipMemsetAsync(W.ctr.p, 0, sizeof(Counters), st));
  HIP_CHECK(hipEventRecord(W.ev[0], st));

  RouteArgs ra{};
  ra.ix = I.view;
  ra.ranges = d_ranges;
  ra.nq = nq;
  ra.method = mcode;
  ra.maxt = maxt;
  ra.qtask_cnt = W.qtask_cnt.p;
  ra.k = k;
  ra.beam = (int32_t)std::min<int64_t>(qp.beam_width, INT32_MAX);
  ra.max_beam = (int32_t)std::min<int64_t>(qp.postfiltering_max_beam, INT32_MAX);
  ra.has_ratio = qp.has_min_query_to_bucket_ratio;
  ra.ratio = qp.min_query_to_bucket_ratio;
  ra.tasks = W.tasks.p;
  ra.graph_list = W.list_a.p;
  ra.graph_count = W.ints.p + I_GRAPH_COUNT;
  ra.heavy_list = W.list_heavy.p;
  ra.heavy_count = W.ints.p + I_HEAVY_COUNT;
  ra.prio_count = W.ints.p + I_PRIO_COUNT;
  ra.heavy_cap = W.big_stride;
  ra.mid_list = W.list_mid.p;
  ra.mid_count = W.ints.p + I_MID_COUNT;
  ra.heavy_ratio = kHeavyRatio;
  ra.risk_count = W.ints.p + I_RISK;
  ra.brute_list = W.list_brute.p;
  ra.brute_count = W.ints.p + I_BRUTE_COUNT;
  ra.spec = spec ? 1 : 0;
  ra.spec_num = 8;
  ra.spec_extra = kSpecExtraLevels;
  const int64_t inkernel_cap = kInKernelBeamCap;
  ra.cap_inkernel = (int32_t)std::max<int64_t>(inkernel_cap, qp.beam_width);
  ra.sub_base0 = (int32_t)(nq * maxt);
  ra.sub_cap = (int32_t)(nq * maxt + sub_slots);
  ra.sub_count = W.ints.p + I_SUB_COUNT;
  // speculative levels beyond the in-kernel cap: searched by "big" workgroups of the first launch (wave 0 owns
  // the LDS of all four waves), beams up to big_cap (the LDS beam must fit; 5792^2 < 2^25 bounds the filter
  // at 32 MiB per workgroup)
  int32_t big_cap = 0;
  if (spec && T.big) {
    const int common = search_lds_bytes_per_wave(I.view.stride, 0);
    const int64_t big_pool = (int64_t)(common + kSearchPoolBytes) * kWavesPerBlock - common;
    big_cap = (int32_t)std::min<int64_t>(big_pool / 8, 5792);
    if (big_cap <= ra.cap_inkernel || (common + kSearchPoolBytes) * kWavesPerBlock > 160 * 1024) big_cap = 0;
  }
  ra.big_cap = big_cap;
  ra.big_list = W.list_big.p;
  ra.big_count = W.ints.p + I_BIG_COUNT;
  ra.big_stride = W.big_stride;
  ra.scan_list = W.list_big.p + 3 * (size_t)W.big_stride;
  ra.scan_count = W.ints.p + I_SCAN_COUNT;
  ra.ctr = W.ctr.p;
  const bool verbose_route = qp.verbose != 0 && (I.host().spec.kind == WANN_KIND_TREE_PREFILTER || I.host().spec.kind == WANN_KIND_TREE_VAMANA ||
                                                   I.host().spec.kind == WANN_KIND_SUPER);
  if (verbose_route) {
    W.vroute.ensure((size_t)nq * kVRouteWords);
    HIP_CHECK(hipMemsetAsync(W.vroute.p, 0, (size_t)nq * kVRouteWords * sizeof(int64_t), st));
    ra.vroute = W.vroute.p;
  }
  if (launch_route(ra, st)) throw HipError(std::string("k_route: ") + launch_last_error());
  // the list sizes come back while the exact scans run: the beam-search launches are sized by them, and skipped
  // altogether for batches without graph tasks (tiny windows) / without levels beyond the in-kernel cap
  const bool sized = I.host().vamana_leaves && qp.beam_width < qp.postfiltering_max_beam;
  if (sized) {
    HIP_CHECK(hipMemcpyAsync(W.h_ints, W.ints.p, kInts * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipEventRecord(W.ev_route, st));
  }

  // The dense path is an optimisation for batches in which queries SHARE windows; on a stream of batches that never form a
  // group its six launches are 30 - 45 us of nothing per batch (a tenth of a 2^-12-window batch).  After two such batches in a
  // row it is only tried every eighth batch until one forms a group again (WANN_DENSE_ALWAYS: every batch).  Results do not
  // depend on it: what the dense path does not take goes through the exact scan.
  bool tried_dense = false;
  if (I.host().spec.kind == WANN_KIND_PREFILTER && nq >= 32 && I.host().spec.dtype == WANN_DTYPE_F32 && T.gemm &&
      (I.dense_idle < 2 || (I.dense_batches & 7) == 0 || T.dense_always)) {
    dense_prefilter(I, T, W, d_queries, nq, k, st);
    tried_dense = true;
  }
  I.dense_batches++;

  const bool may_brute = I.host().spec.kind != WANN_KIND_POSTFILTER && I.host().spec.kind != WANN_KIND_SUPER;
  bool scans_aside = false;
  if (may_brute) {
    BruteArgs ba{};
    ba.ix = I.view;
    ba.queries = d_queries;
    ba.tasks = W.tasks.p;
    ba.list = W.list_brute.p;
    ba.list_count = W.ints.p + I_BRUTE_COUNT;
    ba.cursor = W.ints.p + I_BRUTE_CURSOR;
    ba.k = k;
    ba.out_key = W.out_key.p;
    ba.out_cnt = W.out_cnt.p;
    ba.ctr = W.ctr.p;
    // fenwick / three_split (range_filter_tree.h:297-401,473-540): every query has end scans AND graph searches -- up to two leaf-sized
    // scans per query, a streaming read of ~8 GB per 10 000 queries at n = 10^6 -- and the two touch different task slots.  The scans
    // run beside the searches on a stream of their own (they were 0.7 ms of a 3 ms batch in front of k_search); k_finalize_multi waits
    // for both.  Batches of the one-task methods keep the single stream (their scans are the tiny windows' and mostly absent).
    scans_aside = maxt > 1 && sized;
    hipStream_t scan_st = st;
    if (scans_aside) {
      if (!W.scan_stream) HIP_CHECK(hipStreamCreateWithFlags(&W.scan_stream, hipStreamNonBlocking));
      if (!W.ev_scan) HIP_CHECK(hipEventCreateWithFlags(&W.ev_scan, hipEventDisableTiming));
      scan_st = W.scan_stream;
      HIP_CHECK(hipStreamWaitEvent(scan_st, W.ev_route, 0));  // (recorded behind k_route and this batch's clears)
    }
    if (T.split_scan) {
      const size_t part_cap = (size_t)4 << 20, part_slots = 8192;  // 32 MiB of partial lists
      // (a list is only split while it has far fewer entries than there are waves)
      W.part_key.ensure(part_cap);
      W.part_cnt.ensure(part_slots);
      W.part_done.ensure(part_slots);
      // the kernel leaves zeros behind -- unless a batch ended in an error: 32 KB per batch buy that certainty
      HIP_CHECK(hipMemsetAsync(W.part_done.p, 0, W.part_done.cap * sizeof(int32_t), scan_st));
      ba.part_key = W.part_key.p;
      ba.part_cnt = W.part_cnt.p;
      ba.part_done = W.part_done.p;
      ba.part_cap = (int64_t)part_cap;
      ba.part_slots = (int64_t)part_slots;
    }
    // (as many waves as the chip holds at once -- by the registers: two per SIMD for squared-L2 float rows (two 512-byte rows
    // per lane pair in flight), three for inner-product float rows, five for byte rows -- deal the tickets among themselves)
    const int brute_per_cu = I.view.dtype != WANN_DTYPE_F32 ? 5 : (I.view.metric == 1 ? 3 : 2);
    int blocks = (int)std::min<int64_t>((int64_t)I.num_cus * brute_per_cu, (nq * std::min(maxt, 2) + kWavesPerBlock - 1) / kWavesPerBlock);
    if (launch_brute(ba, blocks, scan_st)) throw HipError(std::string("k_brute: ") + launch_last_error());
    if (scans_aside) HIP_CHECK(hipEventRecord(W.ev_scan, scan_st));
  }

  int rounds = 0, nev = 2;
  std::vector<std::pair<int, int>> timed;  // event index pairs around search launches
  int64_t graph_n = 0, big_n = 0, recovered = 0;
  if (sized) {
    HIP_CHECK(hipEventSynchronize(W.ev_route));
    graph_n = (int64_t)W.h_ints[I_GRAPH_COUNT] + W.h_ints[I_HEAVY_COUNT] + W.h_ints[I_MID_COUNT] + W.h_ints[I_PRIO_COUNT];
    big_n = (int64_t)W.h_ints[I_BIG_COUNT] + W.h_ints[I_BIG_COUNT + 1];
  }
  if (sized && graph_n + big_n > 0) {
    SearchArgs sa{};
    sa.ix = I.view;
    sa.queries = d_queries;
    sa.qid_base = qid_base;
    sa.raw_qids = reinterpret_cast<const long long *>(d_qids);  // (null unless the caller names every query's own id)
    sa.tasks = W.tasks.p;
    sa.k = k;
    sa.limit = qp.limit;
    sa.degree_limit = (int32_t)std::min<int64_t>(qp.degree_limit, INT32_MAX);
    sa.mult = (int32_t)std::min<int64_t>(qp.final_beam_multiply, INT32_MAX);
    sa.max_beam = (int32_t)qp.postfiltering_max_beam;
    sa.pool_bytes = kSearchPoolBytes;
    sa.force_general = (T.force_general || I.view.rs > 64 || verbose_call) ? 1 : 0;  // (wide rows: the register-resident cores take 64 neighbours)
    if (verbose_call) {
      const size_t nt = (size_t)nq * maxt;
      W.vlog.ensure(nt * kVlogCap);
      W.vlog_n.ensure(nt);
      HIP_CHECK(hipMemsetAsync(W.vlog_n.p, 0, nt * sizeof(int32_t), st));
      sa.vlog = W.vlog.p;
      sa.vlog_n = W.vlog_n.p;
      sa.vlog_cap = kVlogCap;
    }
    sa.out_key = W.out_key.p;
    sa.out_cnt = W.out_cnt.p;
    sa.ctr = W.ctr.p;
    sa.final_list = W.list_final.p;
    sa.par_done = W.par_done.p;
    sa.sub_hops = W.sub_hops.p;
    sa.sub_cmps = W.sub_cmps.p;
    sa.next_beam = W.next_beam.p;
    DevBuf<long long> d_trace;  // dev tool: WANN_TASK_TRACE=<file> dumps one line per beam search
    const char *trace_path = T.task_trace.empty() ? nullptr : T.task_trace.c_str();
    const size_t trace_cap = (size_t)1 << 20;
    if (trace_path) {
      d_trace.ensure(1 + 4 * trace_cap);
      HIP_CHECK(hipMemsetAsync(d_trace.p, 0, 8, st));
      sa.trace = d_trace.p;
    }
    // Continuation pollers need the two launches resident together; with launches known to be serialised they are
    // not used at all (continuations then go to the follow-up launch directly).
    const bool use_pollers = T.pollers && (T.force_pollers /* test hook */ || !T.serialized);
    int64_t max_part = 1;
    for (const PartDesc &pd : I.parts) max_part = std::max<int64_t>(max_part, pd.n);
    const int64_t seen_words = ((max_part + 127) / 128) * 4;
    sa.old_general = (T.old_general || I.view.rs > 64 || verbose_call) ? 1 : 0;
    // (idle pollers look for chains that will outgrow their speculated levels)
    const bool scan_on = spec && T.lookahead;
    auto launch = [&](SearchArgs &a, int64_t first_beam, int64_t cap, int64_t items, bool big_lds, int32_t with_big_cap = 0, int32_t deep_pollers = 0,
                      int base_pool = kSearchPoolBytes) {
      RoundCfg rc = config_for(I, T, first_beam, cap, items, big_lds || verbose_call, a.force_general != 0, a.old_general != 0, base_pool);
      big_lds = rc.big_lds;
      a.big_list = nullptr;  // (the one-wave kernel then takes ordinary tickets)
      a.helper = (rc.lc.big == 1 && T.helper) ? kHelpers : 0;
      a.B = (int32_t)first_beam;
      a.cap_inkernel = (int32_t)cap;
      a.pool_bytes = rc.pool_bytes;
      a.g_table = nullptr;
      a.g_beam = nullptr;
      a.npollers = 0;
      a.done_count = nullptr;
      a.big_cap = 0;
      a.yield_for_big = 0;
      a.handoff_beam = 0;
      a.la_count = nullptr;
      a.big_resident = nullptr;
      bool with_big = false;
      SearchArgs big{};
      LaunchCfg big_lc{};
      if (with_big_cap > 0) {
        // companion launch for the speculative levels beyond `cap`: one wave per workgroup, one workgroup per
        // CU, the beam (up to with_big_cap entries) in the LDS, the seen-filter in g_table_big
        with_big = true;
        const int common = search_lds_bytes_per_wave(I.view.stride, 0);
        big = a;
        big_lc.big = a.old_general ? 2 : 1;  // (WANN_OLD_GENERAL: the first-generation core also for the companion's searches)
        big.helper = (big_lc.big == 2 || !T.helper) ? 0 : kHelpers;
        big.cap_inkernel = with_big_cap;
        big.pool_bytes = (common + kSearchPoolBytes) * kWavesPerBlock - common;
        // A companion workgroup of this size shares its CU with an ordinary one (80 KB of LDS and <= 256 registers each).
        // The few pollers of a launch without big items (deep chains) are worth a CU each: they book its whole LDS.
        if (deep_pollers > 0) big.pool_bytes = 150 * 1024 - common;
        big.big_list = W.list_big.p;
        big.big_count = W.ints.p + I_BIG_COUNT;
        big.big_stride = W.big_stride;
        big.big_cursor = W.ints.p + I_BIG_CURSOR;
        big.g_beam = nullptr;
        big.g_table_bits = hash_bits(with_big_cap);
        big_lc.blocks = I.num_cus;  // (cut down to the launch's items + pollers below, once the poller count is known)
        big_lc.waves_per_block = 1;
        // (filter tables / seen bitmaps for one slot per CU whatever the launch's size: a layout that follows the batch would be re-zeroed -- up to 8 GiB -- whenever it changes)
        ensure_filter_scratch(W.g_table_big, W.g_epoch_big, W.g_seen_big, W.g_table_big_layout, big_lc.blocks, big.g_table_bits, seen_words, st);
        big.g_table = W.g_table_big.p;
        big.g_epoch = W.g_epoch_big.p;
        big.g_seen = W.g_seen_big.p;
        big.g_seen_words = seen_words;
        a.yield_for_big = 1;
        a.big_resident = big.big_resident = W.ints.p + I_BIG_RESIDENT;
        a.big_count = W.ints.p + I_BIG_COUNT;
        if (use_pollers) {
          // (companion workgroups share their CUs: pollers are cheap; with the scan on, idle ones look for chains that need a look-ahead)
          a.npollers = big.npollers = deep_pollers > 0 ? deep_pollers : (scan_on ? 32 : 16);
          // (the first npollers companion workgroups never touch the static list: with as many pollers as the launch has workgroups
          // nobody would search the speculated levels, and the tasks would resolve without them -- wrong rows, measured with
          // 256 pollers.  At least half of the workgroups always serve the list; they join the pollers when it is done.)
          a.npollers = big.npollers = std::min<int32_t>(a.npollers, std::max(1, I.num_cus / 2));
          if (deep_pollers > 0) a.handoff_beam = (int32_t)std::max<int64_t>(4 * first_beam, 256);
          // (companion mode hands no in-cap continuation over: measured, profiles/r06_companion_mode_handoff_experiment.txt -- they queue on the pollers)
          if (spec && T.lookahead) {  // look-ahead searches for chains that keep failing (k_search)
            a.la_count = big.la_count = W.ints.p + I_SUB_COUNT;
            a.la_base0 = big.la_base0 = (int32_t)(nq * maxt);
            a.la_cap = big.la_cap = (int32_t)std::min<int64_t>(nq * maxt + sub_slots, INT32_MAX);
            a.la_min_beam = big.la_min_beam = (int32_t)std::max<int64_t>(4 * first_beam, 160);
            a.la_found_max = big.la_found_max = (int32_t)((2 * qp.k + 4) / 5);
            if (deep_pollers == 0 && scan_on) {  // (companion mode: there are speculating tasks)
              big.scan_list = W.list_big.p + 3 * (size_t)W.big_stride;
              big.scan_count = W.ints.p + I_SCAN_COUNT;
              big.scan_min_top = 2560;
              big.scan_num = 16;
            }
            if (T.la_eager) {  // test hook: every chain that fails its second level asks for one
              a.la_min_beam = big.la_min_beam = (int32_t)(2 * first_beam);
              a.la_found_max = big.la_found_max = (int32_t)qp.k;
            }
          }
          big.force_poll_timeout = T.force_poll_timeout ? 1 : 0;  // test hook
          a.big_cap = big.big_cap = with_big_cap;  // (the companion's own chains ask for look-aheads too)
          a.big_count = W.ints.p + I_BIG_COUNT;
          a.dyn_list = big.dyn_list = W.list_big.p + 2 * (size_t)W.big_stride;
          a.dyn_count = big.dyn_count = W.ints.p + I_DYN_COUNT;
          a.dyn_cursor = big.dyn_cursor = W.ints.p + I_DYN_CURSOR;
          a.poll_waiting = big.poll_waiting = W.ints.p + I_POLL_WAITING;
          a.done_count = big.done_count = W.ints.p + I_DONE;
          HIP_CHECK(hipMemsetAsync(a.dyn_list, 0xFF, (size_t)W.big_stride * sizeof(int32_t), st));
        }
        // As many companion workgroups as the launch has work for: its static items + its pollers (round 6; it used to be one
        // workgroup per CU whatever the batch, the surplus leaving at once or idling as extra pollers).  Workgroups that finish
        // their item still join the pollers.  Measured same-box: no change of any fraction's batch time beyond noise
        // (profiles/r06_companion_grid_ab.txt) -- kept because a launch should not ask for 150 KB of LDS 252 times to run 4 pollers.
        {
          const int64_t items = (int64_t)W.h_ints[I_BIG_COUNT] + W.h_ints[I_BIG_COUNT + 1];
          big_lc.blocks = (int)std::max<int64_t>(1, std::min<int64_t>(I.num_cus, items + big.npollers));
        }
      }
      a.g_epoch = nullptr;
      a.g_seen = nullptr;
      if (rc.table_bits) {
        if (rc.lc.big == 1) {  // second-generation core: tagged filter entries + exact seen bitmaps
          ensure_filter_scratch(W.g_table_f, W.g_epoch_f, W.g_seen_f, W.g_table_f_layout, rc.slots, rc.table_bits, seen_words, st);
          a.g_table = W.g_table_f.p;
          a.g_epoch = W.g_epoch_f.p;
          a.g_seen = W.g_seen_f.p;
          a.g_seen_words = seen_words;
        } else if (rc.lc.big == 0) {  // four-wave kernel (third-generation general core): tagged filter entries, no seen bitmaps
          // (the slot count of this launch follows the batch size: the layout key is the table size alone, and the buffers
          // only ever grow -- a fresh allocation is zeroed whole)
          rc.table_bits = ensure_filter_scratch(W.g_table, W.g_epoch, W.g_seen, W.g_table_layout, rc.slots, rc.table_bits, 0, st, /*any_slots=*/true);
          a.g_table = W.g_table.p;
          a.g_epoch = W.g_epoch.p;
        } else {  // legacy one-wave kernel (first-generation general cores): a plain per-slot table, cleared per search
          W.g_table_f.ensure((size_t)rc.slots << rc.table_bits);
          a.g_table = W.g_table_f.p;
          W.g_table_f_layout = -1;  // (plain ids are in it now)
        }
        a.g_table_bits = rc.table_bits;
      }
      if (rc.beam_cap) {
        W.g_beam.ensure((size_t)rc.slots * rc.beam_cap);
        a.g_beam = W.g_beam.p;
        a.g_beam_cap = rc.beam_cap;
      }
      if (nev + 2 > (int)W.ev.size()) throw std::runtime_error("too many search launches in one batch");
      HIP_CHECK(hipEventRecord(W.ev[nev], st));
      if (with_big) {  // first, so that its few workgroups are resident before the ordinary launch fills the CUs
        HIP_CHECK(hipStreamWaitEvent(side, W.ev[nev], 0));
        if (T.profile_phases) {  // (make PROFILE=1 builds: the companion launch's phase cycles, i.e. the long chains UNDER the batch's load)
          I.g_prof.ensure(16);
          HIP_CHECK(hipMemsetAsync(I.g_prof.p, 0, 16 * sizeof(unsigned long long), side));
          big.prof = I.g_prof.p;
        }
        if (launch_search(big, big_lc, side)) throw HipError(std::string("k_search (big): ") + launch_last_error());
        HIP_CHECK(hipEventRecord(W.ev_side, side));
        // The deep-chain pollers book whole CUs, and nothing makes room for them once the ordinary launch has booked every
        // CU for the length of the batch (its workgroups are persistent; the ones that yield free half a CU each).  Which of
        // the two hardware queues reaches the CUs first differed from call to call: on the SIFT-1M 2^-3 batch every second
        // call handed 0 - 3 of its 4 third-level chains over instead of 4 and took 3.19 instead of 2.76 ms.  A one-thread
        // kernel holds this stream until the pollers have started.
        if (deep_pollers > 0)
          if (launch_gate(W.ints.p + I_BIG_RESIDENT, big.npollers, st)) throw HipError(std::string("k_gate: ") + launch_last_error());
      }
      if (launch_search(a, rc.lc, st)) throw HipError(std::string("k_search: ") + launch_last_error());
      if (T.verbose)
        fprintf(stderr, "[wann launch] kind %d: %d workgroups x %d waves (%d per CU by the runtime's occupancy), pool %d B per wave, beams %ld..%ld, %ld items; companion %d (cap %d, pool %d B, pollers %d)\n",
                rc.lc.big, rc.lc.blocks, rc.lc.waves_per_block, search_occupancy(a, rc.lc), a.pool_bytes, (long)first_beam, (long)cap, (long)items, with_big ? big_lc.blocks : 0,
                with_big ? big.cap_inkernel : 0, with_big ? big.pool_bytes : 0, with_big ? big.npollers : 0);
      if (with_big) HIP_CHECK(hipStreamWaitEvent(st, W.ev_side, 0));
      if (with_big && T.profile_phases) {
        unsigned long long h[16];
        HIP_CHECK(hipMemcpyAsync(h, I.g_prof.p, sizeof h, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        fprintf(stderr, "[wann companion phases] select %llu row+probes %llu next+requests %llu slot-test %llu filter %llu next-packet %llu distances %llu "
                        "delta-insert %llu truncation %llu probe-wait %llu flush %llu; hops from the delta list %llu, window hops without a request %llu, with a request but no packet %llu\n", h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8], h[9], h[10], h[11], h[12], h[13]);
      }
      HIP_CHECK(hipEventRecord(W.ev[nev + 1], st));
      timed.emplace_back(nev, nev + 1);
      nev += 2;
      rounds++;
    };
    // launch 1: every task, in-kernel doubling up to kInKernelBeamCap (long tasks first)
    const int64_t b0 = qp.beam_width;
    const int64_t cap1 = std::max<int64_t>(inkernel_cap, b0);
    sa.list = W.list_a.p;
    sa.list_count = W.ints.p + I_GRAPH_COUNT;
    sa.heavy_list = sa.prio_list = W.list_heavy.p;
    sa.heavy_count = W.ints.p + I_HEAVY_COUNT;
    // longest searches first (k_order_heavy) where a launch has levels of several milliseconds: those in the companion launch
    // tell (2^-7 ... 2^-9 of SIFT-1M: 1 ms less per batch; at the wide windows the order of query numbers is as good)
    if (big_n > 0 && W.h_ints[I_HEAVY_COUNT] >= 256) {
      OrderArgs oa{W.tasks.p, W.list_heavy.p, W.list_heavy_ordered.p, W.ints.p + I_HEAVY_COUNT};
      if (launch_order_heavy(oa, st)) throw HipError(std::string("k_order_heavy: ") + launch_last_error());
      sa.heavy_list = W.list_heavy_ordered.p;
    }
    sa.prio_count = W.ints.p + I_PRIO_COUNT;
    sa.heavy_cap = W.big_stride;
    sa.mid_list = W.list_mid.p;
    sa.mid_count = W.ints.p + I_MID_COUNT;
    sa.cursor = W.ints.p + I_CURSOR0;
    sa.next_list = W.list_b.p;
    sa.next_count = W.ints.p + I_NEXT0;
    sa.final_count = W.ints.p + I_FINAL0;
    // the companion launch also runs when there are no levels beyond the cap yet but heavy tasks that may have to
    // double beyond it: its pollers then serve those continuations at once instead of a follow-up launch
    const bool may_continue = W.h_ints[I_RISK] > 0 && use_pollers;
    // A saturated launch without any of that still ends with its few longest chains (a third doubling level started at
    // 1.5 ms of a 2.7 ms bulk runs 1.5 ms there, 0.4 ms alone): a handful of pollers, a CU each, take such chains over.
    const int64_t deep_min = T.deep_min_tasks;
    int32_t deep = 0;
    // (worth a second launch only for a launch of a few milliseconds: tasks x first beam ~ hops)
    // Three workgroups per CU (a leaner LDS pool) where the kernel allows it and no companion workgroup has to share a CU with
    // the ordinary ones (the deep-chain pollers book whole CUs of their own)
    int base_pool = kSearchPoolBytes;
    if (big_n == 0 && !may_continue && cap1 <= kInKernelBeamCap && lean_pool_bytes(I, T) > 0) base_pool = lean_pool_bytes(I, T);
    // How many: with two workgroups per CU (squared-L2 float kernel) four -- 16 cost the SIFT-1M 2^-3 batch 2.5 %; with three
    // (twelve waves share a CU's memory path: a third level takes 2.2 ms there, 1.5 ms on a poller) every third-level chain
    // should find one: 16 (deep-10M-like, eight such chains: 5.3 -> 4.5 ms per batch; 12 ... 32 measure alike).
    if (big_n == 0 && !may_continue && use_pollers && big_cap > 0 && graph_n >= deep_min && graph_n * b0 >= deep_min * 150 &&
        std::max<int64_t>(4 * b0, 256) <= cap1)
      deep = base_pool != kSearchPoolBytes ? 16 : 4;
    launch(sa, b0, cap1, graph_n, false, (big_n > 0 || may_continue || deep > 0) ? big_cap : 0, deep, base_pool);
    HIP_CHECK(hipMemcpyAsync(W.h_ints, W.ints.p, kInts * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    int next_n = W.h_ints[I_NEXT0];
    // continuations handed to the companion launch's pollers that nobody served (the runtime serialised the two
    // launches, or a poller gave up): entries >= 0 of dyn_list.  They join the follow-up launch (next_beam holds
    // the beam each one continues with), so the batch completes with the same rows.
    if (use_pollers && W.h_ints[I_DYN_COUNT] > 0) {
      const int dyn_n = W.h_ints[I_DYN_COUNT];
      std::vector<int32_t> dl((size_t)dyn_n), unserved;
      HIP_CHECK(hipMemcpyAsync(dl.data(), W.list_big.p + 2 * (size_t)W.big_stride, (size_t)dyn_n * 4, hipMemcpyDeviceToHost, st));
      HIP_CHECK(hipStreamSynchronize(st));
      for (int32_t t : dl)
        if (t >= 0 && (int64_t)t < nq * (int64_t)maxt) unserved.push_back(t);  // (entries beyond: look-aheads nobody picked up -- their chains search on)
      if (!unserved.empty()) {
        HIP_CHECK(hipMemcpyAsync(W.list_b.p + next_n, unserved.data(), unserved.size() * 4, hipMemcpyHostToDevice, st));
        next_n += (int)unserved.size();
        HIP_CHECK(hipMemcpyAsync(W.ints.p + I_NEXT0, &next_n, 4, hipMemcpyHostToDevice, st));
        HIP_CHECK(hipStreamSynchronize(st));  // `unserved` / next_n back the uploads
        recovered = (int64_t)unserved.size();
        if (T.verbose) fprintf(stderr, "[wann batch] %d continuations not served by the pollers: re-queued\n", (int)unserved.size());
      }
    }
    // launch 2 (rare): the tasks that must double beyond the cap finish their loop in huge mode
    if (next_n > 0) {
      int64_t nb = b0;
      while (nb <= cap1) nb *= 2;
      SearchArgs sb = sa;
      sb.list = W.list_b.p;
      sb.list_count = W.ints.p + I_NEXT0;
      sb.heavy_list = nullptr;
      sb.heavy_count = nullptr;
      sb.prio_count = nullptr;
      sb.mid_list = nullptr;
      sb.mid_count = nullptr;
      sb.cursor = W.ints.p + I_CURSOR0 + 1;
      sb.next_list = W.list_a.p;  // cannot be used: cap = max_beam
      sb.next_count = W.ints.p + I_NEXT0 + 1;
      sb.start_beam = W.next_beam.p;  // a task resolved from speculative levels may already be past nb
      launch(sb, nb, qp.postfiltering_max_beam, next_n, true);
      HIP_CHECK(hipMemcpyAsync(W.h_ints, W.ints.p, kInts * sizeof(int32_t), hipMemcpyDeviceToHost, st));
      HIP_CHECK(hipStreamSynchronize(st));
    }
    // final re-searches whose beam exceeded the in-kernel cap: every task at its own beam (next_beam), longest
    // first, one wave per workgroup with the beam in the LDS; at most two launches (beams <= 4096, which can
    // run on many more wave slots because their seen-filters are small, then the larger ones)
    const int final_n = W.h_ints[I_FINAL0];
    if (final_n > 0) {
      std::vector<int32_t> fl((size_t)final_n), nbm(W.next_beam.cap);
      HIP_CHECK(hipMemcpyAsync(fl.data(), W.list_final.p, (size_t)final_n * 4, hipMemcpyDeviceToHost, st));
      HIP_CHECK(hipMemcpyAsync(nbm.data(), W.next_beam.p, nbm.size() * 4, hipMemcpyDeviceToHost, st));
      HIP_CHECK(hipStreamSynchronize(st));
      std::sort(fl.begin(), fl.end(), [&](int32_t x, int32_t y) { return nbm[x] != nbm[y] ? nbm[x] > nbm[y] : x < y; });
      const int nlarge = (int)(std::partition_point(fl.begin(), fl.end(), [&](int32_t x) { return nbm[x] > 4096; }) - fl.begin());
      HIP_CHECK(hipMemcpyAsync(W.list_final.p, fl.data(), (size_t)final_n * 4, hipMemcpyHostToDevice, st));
      const int32_t counts[2] = {final_n - nlarge, nlarge};
      HIP_CHECK(hipMemcpyAsync(W.ints.p + I_FINAL0 + 2, counts, 8, hipMemcpyHostToDevice, st));
      for (int g = 0; g < 2; g++) {
        if (counts[g] == 0) continue;
        SearchArgs sf = sa;
        sf.list = W.list_final.p + (g == 0 ? nlarge : 0);
        sf.list_count = W.ints.p + I_FINAL0 + 2 + g;
        sf.heavy_list = nullptr;
        sf.heavy_count = nullptr;
        sf.prio_count = nullptr;
        sf.mid_list = nullptr;
        sf.mid_count = nullptr;
        sf.cursor = W.ints.p + I_CURSOR0 + 2 + g;
        sf.is_final = 1;
        sf.start_beam = W.next_beam.p;
        sf.next_count = W.ints.p + I_NEXT0 + 2;  // unused: a final pass is one search
        sf.final_count = W.ints.p + I_FINAL0 + 1;
        const int64_t gcap = nbm[g == 0 ? fl[nlarge] : fl[0]];  // the group's largest beam
        launch(sf, gcap, gcap, counts[g], true);
      }
      HIP_CHECK(hipStreamSynchronize(st));  // fl / counts back the async uploads
    }
    if (trace_path) {
      HIP_CHECK(hipStreamSynchronize(st));
      std::vector<long long> h(1 + 4 * trace_cap);
      HIP_CHECK(hipMemcpy(h.data(), d_trace.p, h.size() * 8, hipMemcpyDeviceToHost));
      if (FILE *f = fopen(trace_path, "w")) {
        for (long long i = 0; i < std::min<long long>(h[0], (long long)trace_cap); i++)
          fprintf(f, "%lld %lld %lld %lld %lld %lld %lld %lld\n", h[1 + 4 * i] & 0xffffffffll, (h[1 + 4 * i] >> 40) & 1, (h[1 + 4 * i] >> 41) & 1,
                  h[2 + 4 * i] & 0xffffffffll, h[3 + 4 * i], h[4 + 4 * i], (h[1 + 4 * i] >> 32) & 0xff, h[2 + 4 * i] >> 32);  // + found, parent
        fclose(f);
      }
    }
    if (T.verbose) {
      HIP_CHECK(hipStreamSynchronize(st));
      fprintf(stderr, "[wann batch] beam %ld x%ld: next %d final %d big %d+%d heavy %d;", (long)qp.beam_width, (long)qp.final_beam_multiply,
              next_n, final_n, W.h_ints[I_BIG_COUNT], W.h_ints[I_BIG_COUNT + 1], W.h_ints[I_HEAVY_COUNT]);
      for (auto &pr : timed) {
        float t = 0.f;
        HIP_CHECK(hipEventElapsedTime(&t, W.ev[pr.first], W.ev[pr.second]));
        fprintf(stderr, " launch %.2f ms", t);
      }
      fprintf(stderr, "\n");
    }
  }

  if (scans_aside) HIP_CHECK(hipStreamWaitEvent(st, W.ev_scan, 0));
  FinalizeArgs fa{};
  fa.ix = I.view;
  fa.tasks = W.tasks.p;
  fa.maxt = maxt;
  fa.qtask_cnt = W.qtask_cnt.p;
  fa.out_key = W.out_key.p;
  fa.out_cnt = W.out_cnt.p;
  fa.nq = nq;
  fa.k = k;
  fa.decode = I.host().sorted ? 1 : 0;
  // padding ids: tree classes 0 (range_filter_tree.h:90), stand-alone post filter -1
  // (postfilter_vamana.h:212); PrefilterIndex reads past its result there (UB) -> defined as -1
  fa.pad_id = I.host().sorted ? 0u : 0xFFFFFFFFu;
  fa.ids = d_ids;
  fa.dists = d_dists;
  if (launch_finalize(fa, st)) throw HipError(std::string("k_finalize: ") + launch_last_error());
  HIP_CHECK(hipMemcpyAsync(W.h_ctr, W.ctr.p, sizeof(Counters), hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipEventRecord(W.ev[1], st));
  HIP_CHECK(hipStreamSynchronize(st));

  if (W.h_ctr->empty_windows) {
    // range_filter_tree.h:191-203, super_optimized_postfilter_tree.h:173-184: the reference prints this line for every query whose
    // window lies outside the index's label range -- verbose or not -- and returns no neighbours for it
    std::vector<float> hr((size_t)nq * 2);
    float ends[2];
    HIP_CHECK(hipMemcpy(hr.data(), d_ranges, hr.size() * sizeof(float), hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(&ends[0], I.view.labels, sizeof(float), hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(&ends[1], I.view.labels + (I.view.n - 1), sizeof(float), hipMemcpyDeviceToHost));
    // (Four significant digits where the leaves are Vamana graphs: building one makes ParlayANN's timer report, and that sets
    // std::cout's precision to 4 and restores the flags only -- ParlayANN/algorithms/bench/get_time.h:59-68 -- so this is what a
    // reference process prints after its index build; six digits otherwise.)
    const int digits = I.host().vamana_leaves ? 4 : 6;
    for (int64_t q = 0; q < nq; q++)
      if (hr[2 * (size_t)q + 1] < ends[0] || hr[2 * (size_t)q] > ends[1])
        printf("Query range is entirely outside the index range (%.*g, %.*g) index range vs. (%.*g, %.*g) This shouldn't happen but does not directly "
               "impact correctness\n", digits, ends[0], digits, ends[1], digits, hr[2 * (size_t)q], digits, hr[2 * (size_t)q + 1]);
    fflush(stdout);
  }
  if ((verbose_call && W.vlog.p) || verbose_route) {
    static std::mutex dump_mu;  // (WANN_DEVICES: one replica's dump at a time -- whole blocks, not interleaved lines)
    std::lock_guard<std::mutex> dump_lock(dump_mu);
    // the reference's dump (postfilter_vamana.h:155-185 + :230), per query and partition search, in query order
    const size_t nt = (size_t)nq * maxt;
    constexpr int cap_v = kVlogCap;
    std::vector<Task> ht(nt);
    std::vector<int32_t> hq((size_t)nq), hn(nt);
    std::vector<unsigned long long> hv(verbose_call && W.vlog.p ? nt * cap_v : 0);
    HIP_CHECK(hipMemcpy(ht.data(), W.tasks.p, nt * sizeof(Task), hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(hq.data(), W.qtask_cnt.p, (size_t)nq * 4, hipMemcpyDeviceToHost));
    if (!hv.empty()) {  // (a tree with exact-scan leaves has no searches to dump: only the descent's lines)
      HIP_CHECK(hipMemcpy(hn.data(), W.vlog_n.p, nt * 4, hipMemcpyDeviceToHost));
      HIP_CHECK(hipMemcpy(hv.data(), W.vlog.p, hv.size() * 8, hipMemcpyDeviceToHost));
    }
    // the tree classes' own lines around the searches (range_filter_tree.h:452-457, super_optimized_postfilter_tree.h:226-267):
    // what the descent noted, printed before the task it led to.  The two timing lines of the super tree carry the batch's
    // device time per query -- a query has no wall time of its own here.
    std::vector<int64_t> hr;
    if (verbose_route) {
      hr.resize((size_t)nq * kVRouteWords);
      HIP_CHECK(hipMemcpy(hr.data(), W.vroute.p, hr.size() * sizeof(int64_t), hipMemcpyDeviceToHost));
    }
    float batch_ms = 0.f;
    HIP_CHECK(hipEventElapsedTime(&batch_ms, W.ev[0], W.ev[1]));
    const long long per_query_ns = (long long)(batch_ms * 1e6 / (double)std::max<int64_t>(nq, 1));
    for (int64_t q = 0; q < nq; q++) {
      const int64_t *rq = verbose_route ? hr.data() + (size_t)q * kVRouteWords : nullptr;
      const int64_t rwords = rq ? rq[0] : 0;
      bool timed_search = false;
      auto route_lines = [&](int task_index) {  // (entries are in emission order)
        for (int64_t o = 0; o + 7 <= rwords; o += 7) {
          const int64_t *e = rq + 1 + o;
          if (e[1] != task_index) continue;
          if (e[0] == 1) printf("Testing bucket %lld\n", (long long)e[2]);
          else if (e[0] == 2)
            printf("Query range = (%lld,%lld), smallest containing range (size %lld) = (%lld,%lld)\n", (long long)e[2], (long long)e[3], (long long)e[6],
                   (long long)e[4], (long long)e[5]);
          else if (e[0] == 3) {
            printf("Time to find bucket: 0ns\n");
            timed_search = true;
          } else if (e[0] == 4) printf("Query range: %lld %lld\n", (long long)e[2], (long long)e[3]);
          else if (e[0] == 5) printf("Searching bucket: %lld %lld\n", (long long)e[2], (long long)e[3]);
        }
      };
      for (int i = 0; i <= hq[(size_t)q]; i++) {
        route_lines(i);
        if (i == hq[(size_t)q]) break;
        const size_t ti = (size_t)q * maxt + i;
        const Task &t = ht[ti];
        if (t.mode != T_GRAPH || hv.empty()) continue;
        const long long mult = (t.flags & 2) ? 1 : (long long)qp.final_beam_multiply;
        printf("Starting optimized postfiltering, beam size = %lld, k = %lld, final multiply = %lld, n = %d\n", (long long)qp.beam_width,
               (long long)qp.k, mult, I.parts[(size_t)t.part].n);
        long long beam = qp.beam_width, frontier = 0;
        int e = 0;
        const int ne = std::min(hn[ti], cap_v);
        auto rec = [&](int j, long long &b_, long long &m_, long long &f_) {
          const unsigned long long v = hv[ti * cap_v + (size_t)j];
          b_ = (long long)(v >> 42);
          m_ = (long long)((v >> 21) & 0x1fffff);
          f_ = (long long)(v & 0x1fffff);
        };
        while (frontier < qp.k && beam < qp.postfiltering_max_beam && e < ne) {  // :161-172
          long long b_, m_, f_;
          rec(e++, b_, m_, f_);
          printf("Unfiltered return = %lld\n", m_);
          frontier = f_;
          printf("Finished a double, frontier size = %lld, beam size = %lld\n", frontier, beam);
          if (frontier < qp.k) beam *= 2;
        }
        const long long fb = std::min<long long>(beam * mult, qp.postfiltering_max_beam);
        if (fb > beam) {  // :173-181 (the final re-search; should its record be missing -- more searches than records -- the
          if (e < ne) {   // line below still names the beam the reference would)
            long long b_, m_, f_;
            rec(e++, b_, m_, f_);
            printf("Unfiltered return = %lld\n", m_);
            frontier = f_;
          }
          beam = fb;
        }
        printf("Final frontier size = %lld, final beam size %lld\n", frontier, beam);
      }
      if (timed_search) printf("Time to do searcht: %lldns\n", per_query_ns);
    }
    fflush(stdout);
  }
  float ms = 0.f;
  HIP_CHECK(hipEventElapsedTime(&ms, W.ev[0], W.ev[1]));
  last.device_ms = ms;
  double sk = 0;
  for (auto &pr : timed) {
    float t = 0.f;
    HIP_CHECK(hipEventElapsedTime(&t, W.ev[pr.first], W.ev[pr.second]));
    sk += t;
  }
  last.search_kernel_ms = sk;
  last.beam_searches = (int64_t)W.h_ctr->beam_searches;
  last.hops = (int64_t)W.h_ctr->hops;
  last.dist_cmps = (int64_t)W.h_ctr->dist_cmps;
  last.brute_rows = (int64_t)W.h_ctr->brute_rows;
  last.label_reads = (int64_t)W.h_ctr->label_reads;
  last.spec_searches = (int64_t)W.h_ctr->spec_searches;
  last.spec_hops = (int64_t)W.h_ctr->spec_hops;
  last.spec_dist_cmps = (int64_t)W.h_ctr->spec_dist_cmps;
  last.rounds = rounds;
  last.recovered_continuations = recovered;
  last.gemm_queries = (int64_t)W.h_ctr->gemm_queries;
  if (tried_dense) I.dense_idle = W.h_ctr->gemm_queries ? 0 : I.dense_idle.load() + 1;
  last.gemm_unproven = (int64_t)W.h_ctr->gemm_unproven;
  last.gemm_rescued = (int64_t)W.h_ctr->gemm_rescued;
  last.deep_handoffs = (int64_t)W.h_ctr->deep_handoffs;
  last.lookaheads_used = (int64_t)W.h_ctr->lookaheads_used;
  last.big_searches = (int64_t)W.h_ctr->big_searches;
  last.big_hops = (int64_t)W.h_ctr->big_hops;
  last.packet_hops = (int64_t)W.h_ctr->packet_hops;
  last.own_scorings = (int64_t)W.h_ctr->own_scorings;
  last.prefetched_hops = (int64_t)W.h_ctr->prefetched_hops;
  last.poll_timeouts = (int64_t)W.h_ctr->poll_timeouts;
  last.lookaheads_issued = (int64_t)W.h_ctr->lookaheads_issued;
  if (W.h_ctr->unsupported)
    throw std::runtime_error(std::to_string((long long)W.h_ctr->unsupported) +
                             " queries need more than " + std::to_string(maxt) + " partition searches; raise the task slot bound");
}

// Graphs missing from the cache: built on the GPU straight into the adjacency pool (with WANN_HOST_BUILD=1:
// by the host builder and then uploaded -- a test hook, never a fallback).
void upload_part_rows(wann_index &I, const HostPart &P, const PartDesc &pd) {
  const int rs = I.view.rs;
  std::vector<int32_t> stage((size_t)P.n * rs);
  convert_rows(P.g, rs, stage.data());
  HIP_CHECK(hipMemcpy(I.d_graph.p + pd.row_base * rs, stage.data(), stage.size() * 4, hipMemcpyHostToDevice));
}

void build_pending(wann_index &I, std::vector<HostPart *> &pending) {
  HostIndex &H = I.H;
  const BuildSpec &s = H.spec;
  std::vector<GpuBuildTarget> targets;
  size_t pi = 0;
  for (auto &lv : H.levels)
    for (auto &P : lv) {
      if (std::find(pending.begin(), pending.end(), &P) != pending.end()) targets.push_back(GpuBuildTarget{(int32_t)pi, &P});
      pi++;
    }
  // WANN_HOST_BUILD=1 (tests: the host builder as a cross-check) is the only way onto the host builder; a
  // visited list that outgrows its LDS buffer restarts the build ON THE GPU with a larger buffer.
  // (R > 64: the GPU builder's rows are one wave wide -- such graphs are built by the host builder, byte-identical to the
  // reference's like the GPU builder's)
  if (getenv("WANN_HOST_BUILD") != nullptr || s.R > 64) {
    build_pending_on_host(H, pending);
    for (auto &t : targets) upload_part_rows(I, *t.part, I.parts[t.part_index]);
  } else {
    for (int vis_scale = 1;; vis_scale *= 2) {
      try {
        gpu_build_graphs(I.view, I.d_graph.p, I.parts, targets, s.R, s.L, s.alpha, I.num_cus, s.threads, I.own_stream, vis_scale);
        break;
      } catch (std::runtime_error &e) {
        if (std::string(e.what()).find("gpu build overflow: a visited list") == std::string::npos || vis_scale >= 8) throw;
        if (I.tune.verbose) fprintf(stderr, "[wann] %s; restarting the GPU build with a %dx buffer\n", e.what(), 2 * vis_scale);
      }
    }
  }
  save_built_graphs(H, pending, true);
}

std::vector<float> bytes_to_float(int dtype, const void *src, int64_t count) {
  std::vector<float> out((size_t)count);
  if (dtype == WANN_DTYPE_U8) {
    const uint8_t *p = (const uint8_t *)src;
    for (int64_t i = 0; i < count; i++) out[(size_t)i] = (float)p[i];
  } else {
    const int8_t *p = (const int8_t *)src;
    for (int64_t i = 0; i < count; i++) out[(size_t)i] = (float)p[i];
  }
  return out;
}

BuildSpec make_spec(int kind, int metric, int dtype, int64_t n, int64_t d, int32_t cutoff, double split_factor,
                    double shift_factor, const wann_build_params *bp, int threads) {
  BuildSpec s;
  s.kind = kind;
  s.metric = metric;
  s.dtype = dtype;
  s.n = n;
  s.d = d;
  s.cutoff = cutoff;
  s.split_factor = split_factor;
  s.shift_factor = shift_factor;
  s.R = bp ? bp->max_degree : 64;
  s.L = bp ? bp->limit : 500;
  s.alpha = bp ? bp->alpha : 1.175;
  s.cache = (bp && bp->cache_path) ? bp->cache_path : "";
  s.threads = threads;
  return s;
}


}  // namespace wann_host
