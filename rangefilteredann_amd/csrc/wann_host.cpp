// wann_host.cpp -- C ABI (include/wann.h) of the MI355X window-filtered ANN engine: device
// residency of the index, the batch_search driver (routing -> brute scans -> doubling rounds of
// the beam-search kernel -> final re-search -> finalize) and the introspection entry points.
//
// There is no CPU search path in this library: every compute entry point needs a gfx950 device
// and fails loudly otherwise.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <unistd.h>
#include <memory>
#include <condition_variable>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/wann.h"
#include "wann_build.h"
#include "wann_device.h"
#include "wann_gemm_device.h"
#include "wann_gpu_build.h"
#include "wann_hip_util.h"
#include "wann_tuning.h"

#include <rccl/rccl.h>  // types only: librccl.so is opened at first use (wann_batch_search_allgather), never linked

using namespace wann;

namespace {

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

constexpr int kInts = 160;
enum { I_GRAPH_COUNT = 0, I_BRUTE_COUNT = 1, I_BRUTE_CURSOR = 2, I_HEAVY_COUNT = 3, I_SUB_COUNT = 4, I_BIG_COUNT = 5 /* two ints */, I_BIG_CURSOR = 7, I_DYN_COUNT = 140, I_DYN_CURSOR = 141, I_DONE = 142, I_MID_COUNT = 143, I_RISK = 144, I_BIG_RESIDENT = 145, I_POLL_WAITING = 146, I_PRIO_COUNT = 147, I_SCAN_COUNT = 148, I_CURSOR0 = 8, I_NEXT0 = 72, I_FINAL0 = 104 };
constexpr int kMaxRounds = 30;
constexpr int kVlogCap = 24;  // QueryParams::verbose: records per task (a doubling loop from beam 1 to 2^20 is 21 searches + the final one)

struct Workspace {
  DevBuf<Task> tasks;
  DevBuf<int32_t> list_a, list_b, list_final, list_heavy, list_heavy_ordered, list_mid, list_big, list_brute, ints, out_cnt, g_table, g_table_big, qtask_cnt, next_beam, part_cnt, part_done;
  DevBuf<unsigned long long> part_key;
  DevBuf<unsigned long long> out_key, g_beam;
  // wave_beam_search_big: per-slot exact seen bitmaps and filter epochs for the ordinary / follow-up launches
  // (g_table) and for the companion launch (g_table_big); a table and its epochs are zeroed together
  DevBuf<uint32_t> g_seen, g_seen_big, g_seen_f;
  DevBuf<int32_t> g_epoch, g_epoch_big, g_epoch_f, g_table_f;  // (_f: follow-up launches, whose slot layout varies)
  int64_t g_table_layout = -1, g_table_big_layout = -1, g_table_f_layout = -1;  // (slots << 8 | bits) of the last use
  DevBuf<long long> sub_hops, sub_cmps;
  DevBuf<int32_t> par_done;
  DevBuf<Counters> ctr;
  DevBuf<float> q_stage, r_stage, dist_stage;
  DevBuf<uint32_t> id_stage;
  DevBuf<unsigned long long> vlog;  // QueryParams::verbose: per-task records of the doubling loop (SearchArgs::vlog)
  DevBuf<int32_t> vlog_n;
  DevBuf<int32_t> gat_send, gat_recv;  // wann_batch_search_allgather: this replica's [2][cap][k] planes / everybody's [world][2][cap][k]
  int32_t big_stride = 0;
  int32_t *h_ints = nullptr;  // pinned
  Counters *h_ctr = nullptr;  // pinned
  std::vector<hipEvent_t> ev;
  hipEvent_t ev_side = nullptr;   // end of the companion (big) launch on the index's side stream
  hipEvent_t ev_route = nullptr;  // list sizes of k_route are on the host
  ~Workspace() {
    if (h_ints) (void)hipHostFree(h_ints);
    if (h_ctr) (void)hipHostFree(h_ctr);
    for (auto e : ev) (void)hipEventDestroy(e);
    if (ev_side) (void)hipEventDestroy(ev_side);
    if (ev_route) (void)hipEventDestroy(ev_route);
  }
  void ensure(int64_t nq, int k, int maxt, int64_t sub_slots) {
    const size_t nt = (size_t)nq * maxt + (size_t)sub_slots;
    par_done.ensure(nt);
    sub_hops.ensure(nt);
    sub_cmps.ensure(nt);
    tasks.ensure(nt);
    qtask_cnt.ensure(nq);
    list_a.ensure(nt);
    list_b.ensure(nt);
    list_final.ensure(nt);
    list_heavy.ensure(nt);
    list_heavy_ordered.ensure(nt);
    list_mid.ensure(nt);
    list_big.ensure(4 * nt);
    next_beam.ensure(nt);
    big_stride = (int32_t)nt;
    list_brute.ensure(nt);
    ints.ensure(kInts);
    out_cnt.ensure(nt);
    out_key.ensure(nt * k);
    ctr.ensure(1);
    if (!h_ints) HIP_CHECK(hipHostMalloc((void **)&h_ints, kInts * sizeof(int32_t)));
    if (!h_ctr) HIP_CHECK(hipHostMalloc((void **)&h_ctr, sizeof(Counters)));
    if (!ev_side) HIP_CHECK(hipEventCreateWithFlags(&ev_side, hipEventDisableTiming));
    if (!ev_route) HIP_CHECK(hipEventCreateWithFlags(&ev_route, hipEventDisableTiming));
    while (ev.size() < 2 + 4 * kMaxRounds) {
      hipEvent_t e;
      HIP_CHECK(hipEventCreate(&e));
      ev.push_back(e);
    }
  }
};

}  // namespace

struct wann_index {
  HostIndex H;
  // WANN_DEVICES (in-process multi-device mode): further replicas of the device index, one per extra device listed; a replica
  // shares the primary's host index
  HostIndex *Hp = nullptr;
  HostIndex &host() { return Hp ? *Hp : H; }
  const HostIndex &host() const { return Hp ? *Hp : H; }
  std::vector<std::unique_ptr<wann_index>> replicas;
  int device = 0;
  int dtype = WANN_DTYPE_F32;  // element type of the caller's points / host queries (device rows are fp32)
  int num_cus = 256;
  DevBuf<float> d_points, d_labels, d_fv;
  DevBuf<uint32_t> d_decoding;
  DevBuf<int32_t> d_graph, d_fi;
  DevBuf<PartDesc> d_parts;
  DevBuf<int64_t> d_wst_off, d_wst_ptr, d_level_part0, d_level_nb, d_sup_size, d_sup_shift;
  std::vector<PartDesc> parts;
  std::vector<int64_t> level_part0;
  IndexView view{};
  int64_t device_bytes = 0;
  Workspace ws;
  // dense prefilter path (wann_gemm_kernels.hip): |p|^2 per point, computed at first use
  DevBuf<float> d_pnorm2;
  DevBuf<unsigned int> d_pnorm2_max;
  bool have_norms = false;
  // (touched by the blocking calls and by both asynchronous lanes, outside dense_mu for every class but PrefilterIndex)
  std::atomic<int> dense_idle{0};  // batches in a row on which the dense path found no window group (run_batch)
  std::atomic<uint32_t> dense_batches{0};
  DevBuf<GemmGroup> g_groups;
  DevBuf<int32_t> g_gq, g_tile_group, g_tq_group, g_tq_local, g_slot_count, g_slot_group, g_slot_list, g_q_slot, g_q_rank, g_plan;
  DevBuf<unsigned long long> g_slot_key, g_score_used;
  DevBuf<float> g_scores;
  DevBuf<unsigned long long> g_prof;
  hipStream_t own_stream = nullptr;
  hipStream_t side_stream = nullptr;  // companion (big) k_search launches, concurrent with the caller's stream
  wann_counters last{};
  // every WANN_* switch, read when the index is created (wann_tuning.h); run_batch never reads the environment
  // (a call works on a COPY taken under tune_mu -- snapshot_tuning: with WANN_TEST_HOOKS=1 the entry points re-read the record
  // while a lane's worker may still be inside a batch)
  Tuning tune;
  std::mutex tune_mu;
  std::mutex mu;
  std::mutex dense_mu;  // the dense prefilter path's buffers and counters belong to the index: one batch at a time uses them
  // wann_batch_search_device_async: further LANES -- a lane is everything one batch in flight needs (workspace, streams, a
  // worker thread); the blocking calls use the members above
  struct Rccl;                 // librccl.so + one communicator per replica (wann_batch_search_allgather)
  std::unique_ptr<Rccl> rccl;
  struct AsyncLane;
  std::vector<std::unique_ptr<AsyncLane>> lanes;
  std::mutex lanes_mu;   // lanes, next_ticket (held briefly)
  std::mutex gather_mu;  // wann_batch_search_allgather: RCCL set-up, the replicas' send / receive planes and the collective, one call at a time
  std::mutex submit_mu;  // one submission at a time (held while a submitter waits for its lane to fall idle)
  int64_t next_ticket = 0;
  ~wann_index();
};

namespace {

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
void ensure_filter_scratch(DevBuf<int32_t> &table, DevBuf<int32_t> &epoch, DevBuf<uint32_t> &seen, int64_t &layout, int slots,
                           int table_bits, int64_t seen_words, hipStream_t st, bool any_slots = false) {
  const size_t need = (size_t)slots << table_bits;
  // any_slots: a slot's region depends on the table size only (slot << table_bits), so launches with different slot counts
  // share one zeroed table
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
}

struct RoundCfg {
  LaunchCfg lc;
  bool big_lds;      // the one-wave-per-workgroup kernel
  int slots;
  int pool_bytes;    // per-wave LDS pool of this launch
  int table_bits;    // per-slot global seen-filter of 4 << table_bits bytes (0 = none needed)
  int64_t beam_cap;  // per-slot global beam entries (0 = none needed)
};

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

RoundCfg config_for(const wann_index &I, const Tuning &T, int64_t first_beam, int64_t cap, int64_t work_items, bool big_lds = false, bool force_table = false,
                    bool legacy = false, int base_pool = kSearchPoolBytes) {
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
  if (T.blocks_per_cu > 0) blocks_per_cu = std::max(1, std::min(blocks_per_cu, T.blocks_per_cu));  // dev knob
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
  if (T.lean_pool > 0) pool = T.lean_pool;  // dev knob
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
               int64_t qid_base, const char *method, const wann_query_params &qp, uint32_t *d_ids, float *d_dists, hipStream_t st, const Tuning &T) {
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
  ra.prio_count = T.evidence_first ? W.ints.p + I_PRIO_COUNT : nullptr;
  ra.heavy_cap = W.big_stride;
  ra.mid_list = W.list_mid.p;
  ra.mid_count = W.ints.p + I_MID_COUNT;
  ra.heavy_ratio = T.heavy_ratio;
  ra.risk_count = W.ints.p + I_RISK;
  ra.brute_list = W.list_brute.p;
  ra.brute_count = W.ints.p + I_BRUTE_COUNT;
  ra.spec = spec ? 1 : 0;
  ra.spec_num = T.spec_num;
  ra.cap_inkernel = (int32_t)std::max<int64_t>(kInKernelBeamCap, qp.beam_width);
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
    if (T.split_scan) {
      const size_t part_cap = (size_t)4 << 20, part_slots = 8192;  // 32 MiB of partial lists
      // (a list is only split while it has far fewer entries than there are waves)
      W.part_key.ensure(part_cap);
      W.part_cnt.ensure(part_slots);
      W.part_done.ensure(part_slots);
      // the kernel leaves zeros behind -- unless a batch ended in an error: 32 KB per batch buy that certainty
      HIP_CHECK(hipMemsetAsync(W.part_done.p, 0, W.part_done.cap * sizeof(int32_t), st));
      ba.part_key = W.part_key.p;
      ba.part_cnt = W.part_cnt.p;
      ba.part_done = W.part_done.p;
      ba.part_cap = (int64_t)part_cap;
      ba.part_slots = (int64_t)part_slots;
    }
    // (as many waves as the chip holds at once -- by the registers: two per SIMD for squared-L2 float rows (two 512-byte rows
    // per lane pair in flight), three for inner-product float rows, five for byte rows -- deal the tickets among themselves)
    const int brute_per_cu = T.brute_per_cu > 0 ? T.brute_per_cu  // dev knob
                             : (I.view.dtype != WANN_DTYPE_F32 ? 5 : (I.view.metric == 1 ? 3 : 2));
    int blocks = (int)std::min<int64_t>((int64_t)I.num_cus * brute_per_cu, (nq * std::min(maxt, 2) + kWavesPerBlock - 1) / kWavesPerBlock);
    if (launch_brute(ba, blocks, st)) throw HipError(std::string("k_brute: ") + launch_last_error());
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
    sa.search_prio = T.search_prio;  // dev knob
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
    // (idle pollers look for chains that will outgrow their speculated levels: on unless WANN_SCAN=0)
    const bool scan_on = spec && T.scan && T.lookahead;
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
        if (T.big_exclusive >= 0 ? T.big_exclusive != 0 : deep_pollers > 0) big.pool_bytes = 150 * 1024 - common;
        big.big_list = W.list_big.p;
        big.big_count = W.ints.p + I_BIG_COUNT;
        big.big_stride = W.big_stride;
        big.big_cursor = W.ints.p + I_BIG_CURSOR;
        big.g_beam = nullptr;
        big.g_table_bits = hash_bits(with_big_cap);
        big_lc.blocks = I.num_cus;
        big_lc.waves_per_block = 1;
        ensure_filter_scratch(W.g_table_big, W.g_epoch_big, W.g_seen_big, W.g_table_big_layout, big_lc.blocks, big.g_table_bits, seen_words, st);
        big.g_table = W.g_table_big.p;
        big.g_epoch = W.g_epoch_big.p;
        big.g_seen = W.g_seen_big.p;
        big.g_seen_words = seen_words;
        a.yield_for_big = T.yield ? 1 : 0;
        a.big_resident = big.big_resident = W.ints.p + I_BIG_RESIDENT;
        a.big_count = W.ints.p + I_BIG_COUNT;
        if (use_pollers) {
          // (companion workgroups share their CUs: pollers are cheap; with the scan on, idle ones look for chains that need a look-ahead)
          a.npollers = big.npollers = deep_pollers > 0 ? deep_pollers : (T.npollers > 0 ? T.npollers : (scan_on ? 32 : 16));
          if (deep_pollers > 0) a.handoff_beam = (int32_t)std::max<int64_t>(4 * first_beam, 256);
          if (spec && T.lookahead) {  // look-ahead searches for chains that keep failing (k_search)
            a.la_count = big.la_count = W.ints.p + I_SUB_COUNT;
            a.la_base0 = big.la_base0 = (int32_t)(nq * maxt);
            a.la_cap = big.la_cap = (int32_t)std::min<int64_t>(nq * maxt + sub_slots, INT32_MAX);
            a.la_min_beam = big.la_min_beam = (int32_t)std::max<int64_t>(4 * first_beam, 160);
            a.la_found_max = big.la_found_max = (int32_t)((2 * qp.k + 4) / 5);
            if (deep_pollers == 0 && scan_on) {  // (companion mode: there are speculating tasks)
              big.scan_list = W.list_big.p + 3 * (size_t)W.big_stride;
              big.scan_count = W.ints.p + I_SCAN_COUNT;
              big.scan_min_top = T.scan_min_top;
              big.scan_num = T.scan_num;
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
          ensure_filter_scratch(W.g_table, W.g_epoch, W.g_seen, W.g_table_layout, rc.slots, rc.table_bits, 0, st, /*any_slots=*/true);
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
        if (launch_search(big, big_lc, side)) throw HipError(std::string("k_search (big): ") + launch_last_error());
        HIP_CHECK(hipEventRecord(W.ev_side, side));
        // The deep-chain pollers book whole CUs, and nothing makes room for them once the ordinary launch has booked every
        // CU for the length of the batch (its workgroups are persistent; the ones that yield free half a CU each).  Which of
        // the two hardware queues reaches the CUs first differed from call to call: on the SIFT-1M 2^-3 batch every second
        // call handed 0 - 3 of its 4 third-level chains over instead of 4 and took 3.19 instead of 2.76 ms.  A one-thread
        // kernel holds this stream until the pollers have started.
        if (deep_pollers > 0 && T.gate)
          if (launch_gate(W.ints.p + I_BIG_RESIDENT, big.npollers, st)) throw HipError(std::string("k_gate: ") + launch_last_error());
      }
      if (launch_search(a, rc.lc, st)) throw HipError(std::string("k_search: ") + launch_last_error());
      if (T.verbose)
        fprintf(stderr, "[wann launch] kind %d: %d workgroups x %d waves (%d per CU by the runtime's occupancy), pool %d B per wave, beams %ld..%ld, %ld items; companion %d (cap %d, pool %d B, pollers %d)\n",
                rc.lc.big, rc.lc.blocks, rc.lc.waves_per_block, search_occupancy(a, rc.lc), a.pool_bytes, (long)first_beam, (long)cap, (long)items, with_big ? big_lc.blocks : 0,
                with_big ? big.cap_inkernel : 0, with_big ? big.pool_bytes : 0, with_big ? big.npollers : 0);
      if (with_big) HIP_CHECK(hipStreamWaitEvent(st, W.ev_side, 0));
      HIP_CHECK(hipEventRecord(W.ev[nev + 1], st));
      timed.emplace_back(nev, nev + 1);
      nev += 2;
      rounds++;
    };
    // launch 1: every task, in-kernel doubling up to kInKernelBeamCap (long tasks first)
    const int64_t b0 = qp.beam_width;
    const int64_t cap1 = std::max<int64_t>(kInKernelBeamCap, b0);
    sa.list = W.list_a.p;
    sa.list_count = W.ints.p + I_GRAPH_COUNT;
    sa.heavy_list = sa.prio_list = W.list_heavy.p;
    sa.heavy_count = W.ints.p + I_HEAVY_COUNT;
    // longest searches first (k_order_heavy) where a launch has levels of several milliseconds: those in the companion launch
    // tell (2^-7 ... 2^-9 of SIFT-1M: 1 ms less per batch; at the wide windows the order of query numbers is as good)
    if (big_n > 0 && W.h_ints[I_HEAVY_COUNT] >= 256 && T.order) {
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
    if (big_n == 0 && !may_continue && T.lean && cap1 == kInKernelBeamCap && lean_pool_bytes(I, T) > 0) base_pool = lean_pool_bytes(I, T);
    // How many: with two workgroups per CU (squared-L2 float kernel) four -- 16 cost the SIFT-1M 2^-3 batch 2.5 %; with three
    // (twelve waves share a CU's memory path: a third level takes 2.2 ms there, 1.5 ms on a poller) every third-level chain
    // should find one: 16 (deep-10M-like, eight such chains: 5.3 -> 4.5 ms per batch; 12 ... 32 measure alike).
    if (big_n == 0 && !may_continue && use_pollers && big_cap > 0 && graph_n >= deep_min && graph_n * b0 >= deep_min * 150 &&
        T.deep && std::max<int64_t>(4 * b0, 256) <= cap1)
      deep = T.deep_pollers > 0 ? T.deep_pollers : (base_pool != kSearchPoolBytes ? 16 : 4);
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

  if (verbose_call && W.vlog.p) {
    static std::mutex dump_mu;  // (WANN_DEVICES: one replica's dump at a time -- whole blocks, not interleaved lines)
    std::lock_guard<std::mutex> dump_lock(dump_mu);
    // the reference's dump (postfilter_vamana.h:155-185 + :230), per query and partition search, in query order
    const size_t nt = (size_t)nq * maxt;
    constexpr int cap_v = kVlogCap;
    std::vector<Task> ht(nt);
    std::vector<int32_t> hq((size_t)nq), hn(nt);
    std::vector<unsigned long long> hv(nt * cap_v);
    HIP_CHECK(hipMemcpy(ht.data(), W.tasks.p, nt * sizeof(Task), hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(hq.data(), W.qtask_cnt.p, (size_t)nq * 4, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(hn.data(), W.vlog_n.p, nt * 4, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(hv.data(), W.vlog.p, hv.size() * 8, hipMemcpyDeviceToHost));
    for (int64_t q = 0; q < nq; q++)
      for (int i = 0; i < hq[(size_t)q]; i++) {
        const size_t ti = (size_t)q * maxt + i;
        const Task &t = ht[ti];
        if (t.mode != T_GRAPH) continue;
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

}  // namespace

// One batch in flight beside the others: its own workspace, streams and worker thread.  The worker runs the same run_batch
// the blocking call runs (host-side waits included) -- on ITS stream, so the kernels of two consecutive batches share the GPU:
// while batch i's last searches finish, batch i+1 is routed and its first workgroups take the compute units that fall free.
struct wann_index::AsyncLane {
  Workspace ws;
  hipStream_t stream = nullptr, side = nullptr;
  hipEvent_t ready = nullptr;  // the caller's inputs (recorded on the caller's stream at submission)
  wann_counters last{};
  struct Job {
    const float *q, *r;
    int64_t nq, base;
    std::string method;
    wann_query_params qp;
    uint32_t *ids;
    float *dists;
    int64_t ticket;
    Tuning tune;  // the index's switches as they were at submission (the worker never reads the index's record)
  };
  std::thread th;
  std::mutex m;
  std::condition_variable cv;
  bool has_job = false, busy = false, stop = false;
  Job job{};
  int64_t finished = -1;  // ticket of the last finished job; its outcome:
  int rc = WANN_OK;
  std::string err;
  void loop(wann_index *I) {
    for (;;) {
      Job j;
      {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return has_job || stop; });
        if (stop) return;
        j = job;
        has_job = false;
      }
      int code = WANN_OK;
      std::string msg;
      try {
        HIP_CHECK(hipSetDevice(I->device));
        HIP_CHECK(hipStreamWaitEvent(stream, ready, 0));
        run_batch(*I, ws, side, last, j.q, j.r, j.nq, j.base, j.method.c_str(), j.qp, j.ids, j.dists, stream, j.tune);
      } catch (HipError &e) {
        code = WANN_ERR_HIP;
        msg = e.what();
      } catch (std::exception &e) {
        code = WANN_ERR_INVALID;
        msg = e.what();
      }
      {
        std::lock_guard<std::mutex> lk(m);
        rc = code;
        err = msg;
        finished = j.ticket;
        busy = false;
      }
      cv.notify_all();
    }
  }
  ~AsyncLane() {
    {
      std::lock_guard<std::mutex> lk(m);
      stop = true;
    }
    cv.notify_all();
    if (th.joinable()) th.join();
    if (stream) (void)hipStreamDestroy(stream);
    if (side) (void)hipStreamDestroy(side);
    if (ready) (void)hipEventDestroy(ready);
  }
};

// RCCL, opened with dlopen at first use: a host that never gathers on the device never loads it.
struct wann_index::Rccl {
  void *lib = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  std::vector<ncclComm_t> comms;
  void open(const std::vector<int> &devices) {
    lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!lib) throw std::runtime_error(std::string("cannot open librccl.so: ") + dlerror());
    auto sym = [&](const char *n) {
      void *p = dlsym(lib, n);
      if (!p) throw std::runtime_error(std::string("librccl.so lacks ") + n);
      return p;
    };
    CommInitAll = (decltype(CommInitAll))sym("ncclCommInitAll");
    CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
    AllGather = (decltype(AllGather))sym("ncclAllGather");
    GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
    GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
    GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
    comms.assign(devices.size(), nullptr);
    check(CommInitAll(comms.data(), (int)devices.size(), devices.data()), "ncclCommInitAll");
  }
  void check(ncclResult_t r, const char *what) const {
    if (r != ncclSuccess) throw HipError(std::string(what) + ": " + (GetErrorString ? GetErrorString(r) : "RCCL error"));
  }
  ~Rccl() {
    if (CommDestroy)
      for (ncclComm_t c : comms)
        if (c) (void)CommDestroy(c);
    // (the library stays loaded: its teardown at dlclose is not worth the risk at process exit)
  }
};

wann_index::~wann_index() {
  rccl.reset();
  lanes.clear();  // (joins the workers before the streams and buffers they use go away)
  if (own_stream) (void)hipStreamDestroy(own_stream);
  if (side_stream) (void)hipStreamDestroy(side_stream);
}

extern "C" {

int wann_abi_version(void) { return WANN_ABI_VERSION; }
const char *wann_last_error(void) { return g_err.c_str(); }
int wann_device_count(void) { return usable_devices(); }

wann_index *wann_index_create(int kind, int metric, int dtype, const void *points, int64_t n, int64_t d,
                              const float *labels, int32_t cutoff, double split_factor, double shift_factor,
                              const wann_build_params *bp, int device, int build_threads) {
  if (dtype != WANN_DTYPE_F32 && dtype != WANN_DTYPE_U8 && dtype != WANN_DTYPE_I8) {
    fail(WANN_ERR_INVALID, "unknown dtype");
    return nullptr;
  }
  if (kind < 0 || kind > 4 || (metric != 0 && metric != 1) || !points || !labels || n <= 0 || d <= 0) {
    fail(WANN_ERR_INVALID, "invalid argument to wann_index_create");
    return nullptr;
  }
  if (n >= (int64_t)1 << 31) {
    fail(WANN_ERR_UNSUPPORTED, "point sets of 2^31 or more rows are not supported");
    return nullptr;
  }
  // uint8 / int8 point sets (euclidian_point.h:44-60, mips_point.h:44-58: int32 accumulation, cast to float) are kept as
  // BYTE rows on the device and scored with v_dot4 into exact int32 sums: any dimension, a quarter of the vector traffic.
  if (usable_devices() <= device || device < 0) {
    fail(WANN_ERR_NO_DEVICE, "no usable gfx950 device (this library has no CPU search path)");
    return nullptr;
  }
  std::unique_ptr<wann_index> I(new wann_index);
  try {
    I->device = device;
    I->dtype = dtype;
    I->tune = Tuning::from_env();
    I->H.spec = make_spec(kind, metric, dtype, n, d, cutoff, split_factor, shift_factor, bp, build_threads);
    std::vector<HostPart *> pending;
    build_host_index(I->H, points, labels, -1, 0, &pending);
    // WANN_DEVICES=a,b,...: the index is replicated on every listed device and wann_batch_search (host buffers) cuts its batch
    // into contiguous shards, one per replica.  `device` is the primary if it is listed, else the first entry is.
    std::vector<int> extra;
    if (const char *dv = getenv("WANN_DEVICES")) {
      std::vector<int> list;
      for (const char *c = dv; *c;) {
        char *end = nullptr;
        const long v = strtol(c, &end, 10);
        if (end == c) break;
        list.push_back((int)v);
        c = (*end == ',') ? end + 1 : end;
      }
      for (int v : list)
        if (v < 0 || v >= usable_devices()) throw std::runtime_error("WANN_DEVICES names device " + std::to_string(v) + ", which does not exist");
      if (!list.empty()) {
        size_t prim = 0;
        for (size_t i = 0; i < list.size(); i++)
          if (list[i] == device) {
            prim = i;
            break;
          }
        I->device = list[prim];
        for (size_t i = 0; i < list.size(); i++)
          if (i != prim) extra.push_back(list[i]);
      }
    }
    upload_index(*I);
    if (!pending.empty()) build_pending(*I, pending);
    for (int dv : extra) {  // (after the build: the graphs are in the host index by now)
      std::unique_ptr<wann_index> R(new wann_index);
      R->Hp = &I->H;
      R->device = dv;
      R->dtype = dtype;
      R->tune = I->tune;
      upload_index(*R);
      I->replicas.push_back(std::move(R));
    }
    HIP_CHECK(hipSetDevice(I->device));
  } catch (HipError &e) {
    fail(WANN_ERR_HIP, e.what());
    return nullptr;
  } catch (std::exception &e) {
    fail(WANN_ERR_INVALID, e.what());
    return nullptr;
  }
  return I.release();
}

void wann_index_destroy(wann_index *index) { delete index; }

int wann_batch_search_device(wann_index *I, const void *d_queries, const float *d_ranges, int64_t nq,
                             int64_t query_id_base, const char *method, const wann_query_params *qp,
                             uint32_t *d_ids, float *d_dists, void *hip_stream) {
  if (!I || !qp || nq < 0) return fail(WANN_ERR_INVALID, "invalid argument to wann_batch_search_device");
  std::lock_guard<std::mutex> lk(I->mu);
  try {
    // NULL = the HIP default stream: ordered after everything the caller queued on its default stream
    // (torch's current stream unless changed), so freshly produced inputs / recycled output blocks are safe
    hipStream_t st = (hipStream_t)hip_stream;
    const Tuning T = snapshot_tuning(*I);  // (WANN_TEST_HOOKS=1 only: tests flip switches between batches)
    run_batch(*I, I->ws, I->side_stream, I->last, (const float *)d_queries, d_ranges, nq, query_id_base, method, *qp, d_ids, d_dists, st, T);
  } catch (HipError &e) {
    return fail(WANN_ERR_HIP, e.what());
  } catch (std::exception &e) {
    return fail(WANN_ERR_INVALID, e.what());
  }
  return WANN_OK;
}

// Asynchronous form of the device-buffer call (wann.h): tickets are served by kAsyncLanes lanes in turn.
constexpr int kAsyncLanes = 2;

int wann_batch_search_device_async(wann_index *I, const void *d_queries, const float *d_ranges, int64_t nq, int64_t query_id_base,
                                   const char *method, const wann_query_params *qp, uint32_t *d_ids, float *d_dists, void *after_stream,
                                   int64_t *ticket) {
  if (!I || !qp || nq < 0 || !ticket) return fail(WANN_ERR_INVALID, "invalid argument to wann_batch_search_device_async");
  try {
    // One submission at a time; lanes_mu (which wann_wait takes too) is only held to look at / publish the lane table and the
    // ticket counter, never while this call waits for its lane to fall idle.
    std::lock_guard<std::mutex> sub(I->submit_mu);
    HIP_CHECK(hipSetDevice(I->device));
    const Tuning tune = snapshot_tuning(*I);
    int64_t t;
    wann_index::AsyncLane *Lp;
    {
      std::lock_guard<std::mutex> lk(I->lanes_mu);
      if (I->lanes.empty()) {
        // (all lanes or none: a lane table that a failed creation left half filled would be indexed out of bounds by odd tickets)
        std::vector<std::unique_ptr<wann_index::AsyncLane>> fresh;
        for (int l = 0; l < kAsyncLanes; l++) {
          std::unique_ptr<wann_index::AsyncLane> L(new wann_index::AsyncLane);
          int prio_low = 0, prio_high = 0;
          HIP_CHECK(hipDeviceGetStreamPriorityRange(&prio_low, &prio_high));
          HIP_CHECK(hipStreamCreateWithFlags(&L->stream, hipStreamNonBlocking));
          HIP_CHECK(hipStreamCreateWithPriority(&L->side, hipStreamNonBlocking, prio_high));
          HIP_CHECK(hipEventCreateWithFlags(&L->ready, hipEventDisableTiming));
          wann_index::AsyncLane *lp = L.get();
          L->th = std::thread([lp, I] { lp->loop(I); });
          fresh.push_back(std::move(L));
        }
        I->lanes.swap(fresh);
      }
      t = I->next_ticket;
      Lp = I->lanes[(size_t)(t % kAsyncLanes)].get();
    }
    wann_index::AsyncLane &L = *Lp;
    {
      std::unique_lock<std::mutex> ll(L.m);
      L.cv.wait(ll, [&] { return !L.busy; });  // (ticket t - kAsyncLanes has finished; wann_wait it BEFORE submitting this one to see its outcome)
      HIP_CHECK(hipEventRecord(L.ready, (hipStream_t)after_stream));
      L.job = wann_index::AsyncLane::Job{(const float *)d_queries, d_ranges, nq, query_id_base, method ? method : "", *qp, d_ids, d_dists, t, tune};
      L.has_job = true;
      L.busy = true;
    }
    {  // the ticket exists from here on (a submission that failed above took none)
      std::lock_guard<std::mutex> lk(I->lanes_mu);
      I->next_ticket = t + 1;
    }
    L.cv.notify_all();
    *ticket = t;
  } catch (HipError &e) {
    return fail(WANN_ERR_HIP, e.what());
  } catch (std::exception &e) {
    return fail(WANN_ERR_INVALID, e.what());
  }
  return WANN_OK;
}

int wann_wait(wann_index *I, int64_t ticket, wann_counters *out) {
  if (!I || ticket < 0) return fail(WANN_ERR_INVALID, "invalid argument to wann_wait");
  wann_index::AsyncLane *L = nullptr;
  {
    std::lock_guard<std::mutex> lk(I->lanes_mu);
    if (I->lanes.empty() || ticket >= I->next_ticket) return fail(WANN_ERR_INVALID, "wann_wait: no such ticket");
    L = I->lanes[(size_t)(ticket % kAsyncLanes)].get();
  }
  std::unique_lock<std::mutex> ll(L->m);
  L->cv.wait(ll, [&] { return L->finished >= ticket; });
  if (L->finished != ticket) return fail(WANN_ERR_INVALID, "wann_wait: the ticket's lane has served a later ticket since (wait for ticket t before submitting t + 2)");
  if (out) *out = L->last;
  if (L->rc != WANN_OK) return fail(L->rc, L->err);
  return WANN_OK;
}

// Shard `shard` of `world` contiguous shards of an nq-query batch (the cut of wann_batch_search's multi-device mode and of
// wann_batch_search_allgather): first row, row count, and the common plane capacity.
int wann_gather_layout(int64_t nq, int world, int shard, int64_t *lo, int64_t *count, int64_t *cap) {
  if (nq < 0 || world <= 0 || shard < 0 || shard >= world) return fail(WANN_ERR_INVALID, "invalid argument to wann_gather_layout");
  const int64_t base = nq / world, rem = nq % world;
  if (lo) *lo = shard * base + std::min<int64_t>(shard, rem);
  if (count) *count = base + (shard < rem ? 1 : 0);
  if (cap) *cap = base + (rem ? 1 : 0);
  return WANN_OK;
}

// The in-process multi-device call with DEVICE-RESIDENT, gathered result rows: every replica searches its shard into its send
// planes and ONE ncclAllGather (RCCL over xGMI, a communicator per replica, one group call) leaves all shards' planes on every
// replica's device.
int wann_batch_search_allgather(wann_index *I, const void *queries, const float *ranges, int64_t nq, const char *method,
                                const wann_query_params *qp, int32_t **d_planes, int64_t *cap_out) {
  if (!I || !qp || nq < 0 || !d_planes || !cap_out || (nq > 0 && (!queries || !ranges)))
    return fail(WANN_ERR_INVALID, "invalid argument to wann_batch_search_allgather");
  if (qp->k <= 0 || qp->k > 1024) return fail(WANN_ERR_INVALID, "k must be in [1, 1024]");
  const int G = 1 + (int)I->replicas.size();
  std::vector<wann_index *> reps{I};
  for (auto &R : I->replicas) reps.push_back(R.get());
  std::vector<int> devs;
  for (wann_index *T : reps) devs.push_back(T->device);
  for (int a = 0; a < G; a++)
    for (int b = a + 1; b < G; b++)
      if (devs[a] == devs[b]) return fail(WANN_ERR_UNSUPPORTED, "wann_batch_search_allgather needs DISTINCT devices in WANN_DEVICES (one RCCL rank per device)");
  // (one call at a time: the communicators are created lazily, and the planes of a replica's workspace are read by the collective
  // after that replica's own mutex has been released)
  std::lock_guard<std::mutex> gather_lock(I->gather_mu);
  try {
    if (!I->rccl) {
      std::unique_ptr<wann_index::Rccl> r(new wann_index::Rccl);
      r->open(devs);
      I->rccl = std::move(r);
    }
    const int64_t k = qp->k, d = I->H.spec.d, esz = I->dtype == WANN_DTYPE_F32 ? 4 : 1;
    int64_t cap = 0;
    wann_gather_layout(nq, G, 0, nullptr, nullptr, &cap);
    if (cap == 0) cap = 1;
    std::vector<std::thread> threads;
    std::vector<int> codes((size_t)G, WANN_OK);
    std::vector<std::string> errs((size_t)G);
    for (int g = 0; g < G; g++) {
      wann_index *T = reps[(size_t)g];
      int64_t lo = 0, cnt = 0;
      wann_gather_layout(nq, G, g, &lo, &cnt, nullptr);
      threads.emplace_back([=, &codes, &errs] {
        try {
          std::lock_guard<std::mutex> lk(T->mu);
          HIP_CHECK(hipSetDevice(T->device));
          const Tuning tune = snapshot_tuning(*T);
          Workspace &W = T->ws;
          hipStream_t st = T->own_stream;
          W.gat_send.ensure((size_t)(2 * cap * k));
          W.gat_recv.ensure((size_t)((int64_t)G * 2 * cap * k));
          W.q_stage.ensure((size_t)std::max<int64_t>(cnt * d, 1));
          W.r_stage.ensure((size_t)std::max<int64_t>(cnt * 2, 1));
          std::vector<float> qf;
          const void *qsrc = (const char *)queries + lo * d * esz;
          if (cnt && T->dtype != WANN_DTYPE_F32) {
            qf = bytes_to_float(T->dtype, qsrc, cnt * d);
            qsrc = qf.data();
          }
          if (cnt) {
            HIP_CHECK(hipMemcpyAsync(W.q_stage.p, qsrc, (size_t)cnt * d * 4, hipMemcpyHostToDevice, st));
            HIP_CHECK(hipMemcpyAsync(W.r_stage.p, ranges + 2 * lo, (size_t)cnt * 8, hipMemcpyHostToDevice, st));
          }
          int32_t *ids_plane = W.gat_send.p, *dist_plane = W.gat_send.p + cap * k;
          // rows beyond this shard's count: the padding of the reference's result rows (id 0 / FLT_MAX)
          if (cnt < cap) {
            HIP_CHECK(hipMemsetAsync(ids_plane + cnt * k, 0, (size_t)((cap - cnt) * k) * 4, st));
            HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)(dist_plane + cnt * k), 0x7f7fffff, (size_t)((cap - cnt) * k), st));
          }
          run_batch(*T, W, T->side_stream, T->last, W.q_stage.p, W.r_stage.p, cnt, lo, method, *qp, (uint32_t *)ids_plane, (float *)dist_plane, st, tune);
        } catch (HipError &e) {
          codes[(size_t)g] = WANN_ERR_HIP;
          errs[(size_t)g] = e.what();
        } catch (std::exception &e) {
          codes[(size_t)g] = WANN_ERR_INVALID;
          errs[(size_t)g] = e.what();
        }
      });
    }
    for (auto &t : threads) t.join();
    for (int g = 0; g < G; g++)
      if (codes[(size_t)g] != WANN_OK) return fail(codes[(size_t)g], "replica " + std::to_string(g) + ": " + errs[(size_t)g]);
    // one all-gather of the [2][cap][k] planes, all replicas in one group call
    wann_index::Rccl &N = *I->rccl;
    N.check(N.GroupStart(), "ncclGroupStart");
    {
      // (an open group is always closed: a failing call between the two would leave the communicators unusable)
      struct GroupGuard {
        wann_index::Rccl &n;
        bool open = true;
        ~GroupGuard() {
          if (open) (void)n.GroupEnd();
        }
      } guard{N};
      for (int g = 0; g < G; g++) {
        wann_index *T = reps[(size_t)g];
        HIP_CHECK(hipSetDevice(T->device));
        N.check(N.AllGather(T->ws.gat_send.p, T->ws.gat_recv.p, (size_t)(2 * cap * k), ncclInt32, N.comms[(size_t)g], T->own_stream), "ncclAllGather");
      }
      guard.open = false;
      N.check(N.GroupEnd(), "ncclGroupEnd");
    }
    for (int g = 0; g < G; g++) {
      wann_index *T = reps[(size_t)g];
      HIP_CHECK(hipSetDevice(T->device));
      HIP_CHECK(hipStreamSynchronize(T->own_stream));
      d_planes[g] = T->ws.gat_recv.p;
    }
    HIP_CHECK(hipSetDevice(I->device));
    *cap_out = cap;
  } catch (HipError &e) {
    return fail(WANN_ERR_HIP, e.what());
  } catch (std::exception &e) {
    return fail(WANN_ERR_INVALID, e.what());
  }
  return WANN_OK;
}

// Predicted work of every query of a batch (wann.h): the batch is routed on the device without speculative levels and
// k_task_cost prices each query's tasks.  A planning call (one small launch pair and a copy), not part of the search.
int wann_predict_costs(wann_index *I, const float *ranges, int64_t nq, const char *method, const wann_query_params *qp, float *cost) {
  if (!I || !qp || nq < 0 || (nq > 0 && (!ranges || !cost))) return fail(WANN_ERR_INVALID, "invalid argument to wann_predict_costs");
  if (nq == 0) return WANN_OK;
  std::lock_guard<std::mutex> lk(I->mu);
  try {
    HIP_CHECK(hipSetDevice(I->device));
    Workspace &W = I->ws;
    const int mcode = method_code(method);
    const bool tree = I->host().spec.kind == WANN_KIND_TREE_PREFILTER || I->host().spec.kind == WANN_KIND_TREE_VAMANA;
    const bool single = !tree || (mcode == M_OPTIMIZED && !qp->has_min_query_to_bucket_ratio && I->host().spec.split_factor <= 4);
    const int maxt = single ? 1 : 96;
    const int k = (int)std::max<int64_t>(1, std::min<int64_t>(qp->k, 1024));
    W.ensure(nq, k, maxt, 0);
    W.r_stage.ensure((size_t)nq * 2);
    W.dist_stage.ensure((size_t)nq);
    const Tuning tune = snapshot_tuning(*I);
    hipStream_t st = I->own_stream;
    HIP_CHECK(hipMemcpyAsync(W.r_stage.p, ranges, (size_t)nq * 8, hipMemcpyHostToDevice, st));
    HIP_CHECK(hipMemsetAsync(W.ints.p, 0, kInts * sizeof(int32_t), st));
    HIP_CHECK(hipMemsetAsync(W.ctr.p, 0, sizeof(Counters), st));
    RouteArgs ra{};
    ra.ix = I->view;
    ra.ranges = W.r_stage.p;
    ra.nq = nq;
    ra.method = mcode;
    ra.maxt = maxt;
    ra.qtask_cnt = W.qtask_cnt.p;
    ra.k = k;
    ra.beam = (int32_t)std::min<int64_t>(std::max<int64_t>(qp->beam_width, 1), INT32_MAX);
    ra.max_beam = (int32_t)std::min<int64_t>(qp->postfiltering_max_beam, INT32_MAX);
    ra.has_ratio = qp->has_min_query_to_bucket_ratio;
    ra.ratio = qp->min_query_to_bucket_ratio;
    ra.tasks = W.tasks.p;
    ra.graph_list = W.list_a.p;
    ra.graph_count = W.ints.p + I_GRAPH_COUNT;
    ra.heavy_list = W.list_heavy.p;
    ra.heavy_count = W.ints.p + I_HEAVY_COUNT;
    ra.heavy_cap = W.big_stride;
    ra.mid_list = W.list_mid.p;
    ra.mid_count = W.ints.p + I_MID_COUNT;
    ra.heavy_ratio = tune.heavy_ratio;
    ra.risk_count = W.ints.p + I_RISK;
    ra.brute_list = W.list_brute.p;
    ra.brute_count = W.ints.p + I_BRUTE_COUNT;
    ra.spec = 0;  // (plain tasks only: each carries its window's size)
    ra.spec_num = tune.spec_num;
    ra.cap_inkernel = (int32_t)std::max<int64_t>(kInKernelBeamCap, qp->beam_width);
    ra.sub_base0 = ra.sub_cap = (int32_t)(nq * maxt);
    ra.sub_count = W.ints.p + I_SUB_COUNT;
    ra.big_list = W.list_big.p;
    ra.big_count = W.ints.p + I_BIG_COUNT;
    ra.big_stride = W.big_stride;
    ra.ctr = W.ctr.p;
    if (launch_route(ra, st)) throw HipError(std::string("k_route: ") + launch_last_error());
    CostArgs ca{};
    ca.tasks = W.tasks.p;
    ca.qtask_cnt = W.qtask_cnt.p;
    ca.parts = I->view.parts;
    ca.nq = nq;
    ca.maxt = maxt;
    ca.k = k;
    ca.beam = ra.beam;
    ca.max_beam = ra.max_beam;
    ca.mult = (int32_t)std::min<int64_t>(std::max<int64_t>(qp->final_beam_multiply, 1), INT32_MAX);
    ca.cost = W.dist_stage.p;
    if (launch_task_cost(ca, st)) throw HipError(std::string("k_task_cost: ") + launch_last_error());
    HIP_CHECK(hipMemcpyAsync(cost, W.dist_stage.p, (size_t)nq * 4, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
  } catch (HipError &e) {
    return fail(WANN_ERR_HIP, e.what());
  } catch (std::exception &e) {
    return fail(WANN_ERR_INVALID, e.what());
  }
  return WANN_OK;
}

namespace {
// one replica's share of a host-buffer call: stage, search (queries keep their global numbers), copy back
void search_host_one(wann_index &T, const void *queries, const float *ranges, int64_t nq, int64_t qid_base, const char *method,
                     const wann_query_params &qp, uint32_t *ids, float *dists) {
  std::lock_guard<std::mutex> lk(T.mu);
  HIP_CHECK(hipSetDevice(T.device));
  const Tuning tune = snapshot_tuning(T);  // (WANN_TEST_HOOKS=1 only: re-read)
  Workspace &W = T.ws;
  const int64_t d = T.host().spec.d;
  if (qp.k <= 0 || qp.k > 1024) throw std::runtime_error("k must be in [1, 1024]");
  W.q_stage.ensure((size_t)nq * d);
  W.r_stage.ensure((size_t)nq * 2);
  W.id_stage.ensure((size_t)nq * qp.k);
  W.dist_stage.ensure((size_t)nq * qp.k);
  hipStream_t st = T.own_stream;
  std::vector<float> qf;  // host queries arrive in the index's element type
  if (nq && T.dtype != WANN_DTYPE_F32) {
    qf = bytes_to_float(T.dtype, queries, nq * d);
    queries = qf.data();
  }
  if (nq) {
    HIP_CHECK(hipMemcpyAsync(W.q_stage.p, queries, (size_t)nq * d * 4, hipMemcpyHostToDevice, st));
    HIP_CHECK(hipMemcpyAsync(W.r_stage.p, ranges, (size_t)nq * 8, hipMemcpyHostToDevice, st));
  }
  run_batch(T, W, T.side_stream, T.last, W.q_stage.p, W.r_stage.p, nq, qid_base, method, qp, W.id_stage.p, W.dist_stage.p, st, tune);
  if (nq) {
    HIP_CHECK(hipMemcpyAsync(ids, W.id_stage.p, (size_t)nq * qp.k * 4, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipMemcpyAsync(dists, W.dist_stage.p, (size_t)nq * qp.k * 4, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
  }
}
}  // namespace

int wann_batch_search(wann_index *I, const void *queries, const float *ranges, int64_t nq, const char *method,
                      const wann_query_params *qp, uint32_t *ids, float *dists) {
  if (!I || !qp || nq < 0 || (nq > 0 && (!queries || !ranges || !ids || !dists)))
    return fail(WANN_ERR_INVALID, "invalid argument to wann_batch_search");
  const int G = 1 + (int)I->replicas.size();
  if (G == 1 || nq < G) {
    try {
      search_host_one(*I, queries, ranges, nq, 0, method, *qp, ids, dists);
    } catch (HipError &e) {
      return fail(WANN_ERR_HIP, e.what());
    } catch (std::exception &e) {
      return fail(WANN_ERR_INVALID, e.what());
    }
    return WANN_OK;
  }
  // In-process multi-device mode (WANN_DEVICES): contiguous shards that keep their global query numbers (the reference uses
  // a query's row number as its own id, range_filter_tree.h:62-96 + beamSearch.h:128), one host thread and one stream per
  // replica, rows land in the caller's arrays.
  const int64_t d = I->H.spec.d, esz = I->dtype == WANN_DTYPE_F32 ? 4 : 1;
  std::vector<std::thread> threads;
  std::vector<int> codes((size_t)G, WANN_OK);
  std::vector<std::string> errs((size_t)G);
  for (int g = 0; g < G; g++) {
    const int64_t base = nq / G, rem = nq % G;
    const int64_t lo = g * base + std::min<int64_t>(g, rem), cnt = base + (g < rem ? 1 : 0);
    wann_index *T = g == 0 ? I : I->replicas[(size_t)g - 1].get();
    threads.emplace_back([=, &codes, &errs] {
      try {
        search_host_one(*T, (const char *)queries + lo * d * esz, ranges + 2 * lo, cnt, lo, method, *qp, ids + lo * qp->k, dists + lo * qp->k);
      } catch (HipError &e) {
        codes[(size_t)g] = WANN_ERR_HIP;
        errs[(size_t)g] = e.what();
      } catch (std::exception &e) {
        codes[(size_t)g] = WANN_ERR_INVALID;
        errs[(size_t)g] = e.what();
      }
    });
  }
  for (auto &t : threads) t.join();
  (void)hipSetDevice(I->device);
  for (int g = 0; g < G; g++)
    if (codes[(size_t)g] != WANN_OK) return fail(codes[(size_t)g], "replica " + std::to_string(g) + ": " + errs[(size_t)g]);
  // counters of the call: work summed over the replicas, times of the slowest one
  wann_counters sum = I->last;
  for (auto &R : I->replicas) {
    const wann_counters &c = R->last;
    sum.beam_searches += c.beam_searches;
    sum.hops += c.hops;
    sum.dist_cmps += c.dist_cmps;
    sum.brute_rows += c.brute_rows;
    sum.label_reads += c.label_reads;
    sum.rounds = std::max(sum.rounds, c.rounds);
    sum.spec_searches += c.spec_searches;
    sum.spec_hops += c.spec_hops;
    sum.spec_dist_cmps += c.spec_dist_cmps;
    sum.gemm_queries += c.gemm_queries;
    sum.gemm_unproven += c.gemm_unproven;
    sum.gemm_rescued += c.gemm_rescued;
    sum.recovered_continuations += c.recovered_continuations;
    sum.deep_handoffs += c.deep_handoffs;
    sum.lookaheads_used += c.lookaheads_used;
    sum.lookaheads_issued += c.lookaheads_issued;
    sum.big_searches += c.big_searches;
    sum.big_hops += c.big_hops;
    sum.packet_hops += c.packet_hops;
    sum.own_scorings += c.own_scorings;
    sum.prefetched_hops += c.prefetched_hops;
    sum.poll_timeouts += c.poll_timeouts;
    sum.device_ms = std::max(sum.device_ms, c.device_ms);
    sum.search_kernel_ms = std::max(sum.search_kernel_ms, c.search_kernel_ms);
  }
  I->last = sum;
  return WANN_OK;
}

int wann_get_counters(const wann_index *I, wann_counters *out) {
  if (!I || !out) return fail(WANN_ERR_INVALID, "null argument");
  *out = I->last;
  return WANN_OK;
}

int64_t wann_num_points(const wann_index *I) { return I ? I->H.spec.n : -1; }
int64_t wann_dim(const wann_index *I) { return I ? I->H.spec.d : -1; }
int64_t wann_num_levels(const wann_index *I) { return I ? (int64_t)I->H.levels.size() : -1; }
int64_t wann_level_size(const wann_index *I, int64_t level) {
  if (!I || level < 0 || level >= (int64_t)I->H.levels.size()) return -1;
  return (int64_t)I->H.levels[level].size();
}
int wann_partition_range(const wann_index *I, int64_t level, int64_t idx, int64_t *start, int64_t *end) {
  if (!I || level < 0 || level >= (int64_t)I->H.levels.size() || idx < 0 || idx >= (int64_t)I->H.levels[level].size())
    return fail(WANN_ERR_INVALID, "partition out of range");
  const HostPart &P = I->H.levels[level][idx];
  *start = P.start;
  *end = P.start + P.n;
  return WANN_OK;
}
int wann_partition_graph(const wann_index *I, int64_t level, int64_t idx, int32_t *rows, int64_t cap_rows, int64_t max_degree) {
  if (!I || !rows || level < 0 || level >= (int64_t)I->H.levels.size() || idx < 0 || idx >= (int64_t)I->H.levels[level].size())
    return fail(WANN_ERR_INVALID, "partition out of range");
  const HostPart &P = I->H.levels[level][idx];
  if (P.g.n != P.n || cap_rows < P.n) return fail(WANN_ERR_INVALID, "no graph / buffer too small");
  if (max_degree != (int64_t)P.g.maxdeg)
    return fail(WANN_ERR_INVALID, "max_degree " + std::to_string((long long)max_degree) + " does not match the index's R = " +
                                      std::to_string((long long)P.g.maxdeg) + " (rows are R+1 ints wide)");
  memcpy(rows, P.g.rows.data(), (size_t)P.n * (size_t)(P.g.maxdeg + 1) * 4);
  return WANN_OK;
}
int64_t wann_max_degree(const wann_index *I) { return I ? I->H.spec.R : -1; }
int64_t wann_device_bytes(const wann_index *I) { return I ? I->device_bytes : -1; }
int wann_num_replicas(const wann_index *I) { return I ? 1 + (int)I->replicas.size() : -1; }

int wann_build_cache_shard(int kind, int metric, int dtype, const void *points, int64_t n, int64_t d,
                           const float *labels, int32_t cutoff, double split_factor, double shift_factor,
                           const wann_build_params *bp, int shard, int nshards, int build_threads) {
  if (dtype != WANN_DTYPE_F32 && dtype != WANN_DTYPE_U8 && dtype != WANN_DTYPE_I8) return fail(WANN_ERR_INVALID, "unknown dtype");
  if (!bp || !bp->cache_path || !*bp->cache_path) return fail(WANN_ERR_INVALID, "cache_path required");
  if (nshards <= 0 || shard < 0 || shard >= nshards) return fail(WANN_ERR_INVALID, "bad shard");
  try {
    HostIndex H;
    H.spec = make_spec(kind, metric, dtype, n, d, cutoff, split_factor, shift_factor, bp, build_threads);
    build_host_index(H, points, labels, shard, nshards);
  } catch (std::exception &e) {
    return fail(WANN_ERR_INVALID, e.what());
  }
  return WANN_OK;
}

// One graph over one contiguous slice of a point set, resident on the device: the object behind wann_raw_beam_search
// (tests, micro-benchmarks) and behind the unfiltered VamanaIndex API.
struct RawGraph {
  wann_index I;  // scratch object: config_for / device properties
  DevBuf<float> d_pts;
  DevBuf<int32_t> d_rows;
  DevBuf<PartDesc> d_parts;
  int64_t n = 0, d = 0, subset_n = 0;
  int32_t maxdeg = 0;
  // per-call buffers, kept across calls (a VamanaIndex answers batch after batch)
  DevBuf<float> d_q, d_rd;
  DevBuf<int32_t> d_list, d_ints, d_rid, d_rsz, g_table, g_epoch;
  DevBuf<Task> d_tasks;
  DevBuf<long long> d_hops, d_cmps, d_qids;
  DevBuf<Counters> d_ctr;
  DevBuf<unsigned long long> g_beam, d_prof;
  DevBuf<uint32_t> g_seen;
  int64_t layout = -1;
  // points: (n, d) rows of `dtype` elements (float32, or uint8 / int8 bytes: stored as byte rows)
  void load(int device, int metric, const void *points, int64_t n_, int64_t d_, const int32_t *graph_rows, int64_t maxdeg_,
            int64_t subset_start, int64_t subset_n_, int dtype = WANN_DTYPE_F32) {
    HIP_CHECK(hipSetDevice(device));
    I.device = device;
    I.tune = Tuning::from_env();
    hipDeviceProp_t prop;
    HIP_CHECK(hipGetDeviceProperties(&prop, device));
    I.num_cus = prop.multiProcessorCount;
    n = n_;
    d = d_;
    subset_n = subset_n_;
    maxdeg = (int32_t)maxdeg_;
    const int64_t esz = dtype == WANN_DTYPE_F32 ? 4 : 1;
    const int64_t stride = ((d * esz + 63) / 64) * 16;  // 32-bit words per row
    std::vector<float> pts((size_t)n * stride, 0.f);
    for (int64_t i = 0; i < n; i++) memcpy(pts.data() + i * stride, (const char *)points + i * d * esz, (size_t)(d * esz));
    const int rs = (int)(((maxdeg_ + 15) / 16) * 16);
    HostGraph g;
    g.n = subset_n;
    g.maxdeg = (int32_t)maxdeg_;
    g.rows.assign(graph_rows, graph_rows + (size_t)subset_n * (maxdeg_ + 1));
    std::vector<int32_t> rows((size_t)subset_n * rs);
    convert_rows(g, rs, rows.data());
    d_pts.upload(pts);
    d_rows.upload(rows);
    std::vector<PartDesc> parts{{0, (int32_t)subset_start, (int32_t)subset_n}};
    d_parts.upload(parts);
    I.view.points = d_pts.p;
    I.view.graph = d_rows.p;
    I.view.parts = d_parts.p;
    I.view.labels = d_pts.p;  // unused in raw mode
    I.view.n = n;
    I.view.d = (int32_t)d;
    I.view.stride = (int32_t)stride;
    I.view.rs = rs;
    I.view.maxdeg = (int32_t)maxdeg_;
    I.view.metric = metric;
    I.view.dtype = dtype;
  }
  // one beam search per query (host buffers); cut_k > 0: the k / cut step of beamSearch.h:159-167 (first-generation core)
  void search(const float *queries, int64_t nq, const int64_t *query_ids, int64_t beam, int64_t limit, int64_t degree_limit,
              int64_t cut_k, double cut, int32_t *out_ids, float *out_dists, int32_t *out_sizes, int64_t *out_hops, int64_t *out_dist_cmps) {
    HIP_CHECK(hipSetDevice(I.device));
    std::vector<float> qv(queries, queries + (size_t)nq * d);
    d_q.upload(qv);
    std::vector<Task> tasks((size_t)nq);
    std::vector<int32_t> list((size_t)nq);
    std::vector<long long> qids((size_t)nq);
    for (int64_t i = 0; i < nq; i++) {
      tasks[i] = Task{(int32_t)i, T_GRAPH, 0, 0, 0, 0, 0.f, 0.f};
      list[i] = (int32_t)i;
      qids[i] = query_ids ? query_ids[i] : i;
    }
    d_tasks.upload(tasks);
    d_list.upload(list);
    d_qids.upload(qids);
    std::vector<int32_t> ints{(int32_t)nq, 0, 0, 0};
    d_ints.upload(ints);
    d_rid.ensure((size_t)nq * beam);
    d_rd.ensure((size_t)nq * beam);
    d_rsz.ensure(nq);
    d_hops.ensure(nq);
    d_cmps.ensure(nq);
    d_ctr.ensure(1);
    HIP_CHECK(hipMemset(d_ctr.p, 0, sizeof(Counters)));
    const bool with_cut = cut_k > 0;
    if (I.tune.hooks_live) I.tune = Tuning::from_env();  // (tests flip the core switches between calls on one VamanaIndex)
    const Tuning &T = I.tune;
    const bool wide = I.view.rs > 64;
    const bool old_general = T.old_general || with_cut || wide, force_general = T.force_general || with_cut || wide;
    // (dev / test switches: the large-LDS one-wave configuration; the first-generation cores live in that kernel only)
    // (dev: WANN_LEAN_POOL under WANN_TEST_HOOKS=1 gives the raw search the leaner per-wave pool too -- three workgroups per CU)
    const int raw_pool = (T.hooks_live && T.lean_pool > 0) ? T.lean_pool : kSearchPoolBytes;
    RoundCfg rc = config_for(I, T, beam, beam, nq, T.raw_big_lds || old_general, force_general, old_general, raw_pool);
    SearchArgs sa{};
    sa.ix = I.view;
    sa.queries = d_q.p;
    sa.tasks = d_tasks.p;
    sa.list = d_list.p;
    sa.list_count = d_ints.p;
    sa.cursor = d_ints.p + 1;
    sa.B = (int32_t)beam;
    sa.cap_inkernel = (int32_t)beam;
    sa.max_beam = INT32_MAX;
    sa.mult = 1;
    sa.pool_bytes = rc.pool_bytes;
    sa.force_general = force_general ? 1 : 0;
    sa.k = 1;
    sa.limit = limit;
    sa.degree_limit = (int32_t)std::min<int64_t>(degree_limit, INT32_MAX);
    sa.ctr = d_ctr.p;
    sa.raw = 1;
    sa.raw_ids = d_rid.p;
    sa.raw_dists = d_rd.p;
    sa.raw_sizes = d_rsz.p;
    sa.raw_hops = d_hops.p;
    sa.raw_cmps = d_cmps.p;
    sa.raw_qids = d_qids.p;
    sa.cut_k = (int32_t)cut_k;
    sa.cut = cut;
    sa.old_general = old_general ? 1 : 0;
    sa.helper = (rc.lc.big == 1 && T.helper) ? kHelpers : 0;
    if (rc.table_bits) {
      const int64_t seen_words = ((subset_n + 127) / 128) * 4;
      ensure_filter_scratch(g_table, g_epoch, g_seen, layout, rc.slots, rc.table_bits, seen_words, nullptr);
      sa.g_table = g_table.p;
      sa.g_table_bits = rc.table_bits;
      sa.g_epoch = g_epoch.p;
      sa.g_seen = g_seen.p;
      sa.g_seen_words = seen_words;
    }
    if (rc.beam_cap) {
      sa.g_beam_cap = rc.beam_cap;
      g_beam.ensure((size_t)rc.slots * sa.g_beam_cap);
      sa.g_beam = g_beam.p;
    }
    const bool prof = T.profile_phases;
    if (prof) {
      d_prof.ensure(16);
      HIP_CHECK(hipMemset(d_prof.p, 0, 16 * sizeof(unsigned long long)));
      sa.prof = d_prof.p;
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const bool verbose = T.verbose;
    if (verbose) {
      HIP_CHECK(hipEventCreate(&e0));
      HIP_CHECK(hipEventCreate(&e1));
      HIP_CHECK(hipEventRecord(e0, nullptr));
    }
    if (launch_search(sa, rc.lc, nullptr)) throw HipError(std::string("k_search: ") + launch_last_error());
    if (verbose) HIP_CHECK(hipEventRecord(e1, nullptr));
    HIP_CHECK(hipDeviceSynchronize());
    if (verbose) {
      float ms = 0.f;
      HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
      fprintf(stderr, "[wann raw] beam %ld nq %ld kernel kind %d blocks %d (%d per CU by the runtime's occupancy): %.3f ms\n", (long)beam, (long)nq, rc.lc.big, rc.lc.blocks,
              search_occupancy(sa, rc.lc), ms);
      (void)hipEventDestroy(e0);
      (void)hipEventDestroy(e1);
    }
    if (prof) {
      unsigned long long h[16];
      HIP_CHECK(hipMemcpy(h, d_prof.p, sizeof h, hipMemcpyDeviceToHost));
      fprintf(stderr, "[wann phases] beam=%ld nq=%ld cycles: row %llu filter %llu dist %llu merge %llu next %llu (built with make PROFILE=1?)\n", (long)beam,
              (long)nq, h[0], h[1], h[2], h[3], h[4]);
      fprintf(stderr, "[wann phases 5..8] %llu %llu %llu %llu (second-generation core: select / row+probes / next+requests / slot test / filter / "
                      "next packet / distances / delta insert / truncation); probe wait %llu, flush %llu\n", h[5], h[6], h[7], h[8], h[9], h[10]);
    }
    HIP_CHECK(hipMemcpy(out_ids, d_rid.p, (size_t)nq * beam * 4, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(out_dists, d_rd.p, (size_t)nq * beam * 4, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(out_sizes, d_rsz.p, (size_t)nq * 4, hipMemcpyDeviceToHost));
    std::vector<long long> hh((size_t)nq), cc((size_t)nq);
    HIP_CHECK(hipMemcpy(hh.data(), d_hops.p, (size_t)nq * 8, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(cc.data(), d_cmps.p, (size_t)nq * 8, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < nq; i++) {
      if (out_hops) out_hops[i] = hh[i];
      if (out_dist_cmps) out_dist_cmps[i] = cc[i];
    }
  }
};

int wann_raw_beam_search(int metric, const float *points, int64_t n, int64_t d, const int32_t *graph_rows,
                         int64_t maxdeg, int64_t subset_start, int64_t subset_n, const float *queries, int64_t nq,
                         const int64_t *query_ids, int64_t beam, int64_t limit, int64_t degree_limit,
                         int32_t *out_ids, float *out_dists, int32_t *out_sizes, int64_t *out_hops,
                         int64_t *out_dist_cmps, int device) {
  if (usable_devices() <= device || device < 0)
    return fail(WANN_ERR_NO_DEVICE, "no usable gfx950 device (this library has no CPU search path)");
  if (maxdeg > WANN_MAX_DEGREE) return fail(WANN_ERR_UNSUPPORTED, "max_degree > 128 is not supported");
  try {
    RawGraph G;
    G.load(device, metric, points, n, d, graph_rows, maxdeg, subset_start, subset_n);
    G.search(queries, nq, query_ids, beam, limit, degree_limit, 0, 0.0, out_ids, out_dists, out_sizes, out_hops, out_dist_cmps);
  } catch (HipError &e) {
    return fail(WANN_ERR_HIP, e.what());
  } catch (std::exception &e) {
    return fail(WANN_ERR_INVALID, e.what());
  }
  return WANN_OK;
}

// ---- unfiltered VamanaIndex (ParlayANN/python/vamana_index.cpp:42-76, ParlayANN/python/builder.cpp) ----------------
namespace {
// point file: uint32 n, uint32 d, then n * d elements (point_range.h:63-93)
void read_point_file(const char *path, int dtype, std::vector<float> &out, int64_t &n, int64_t &d, std::vector<unsigned char> *raw_out = nullptr) {
  FILE *f = fopen(path, "rb");
  if (!f) throw std::runtime_error(std::string("cannot open point file ") + path);
  uint32_t head[2];
  if (fread(head, 4, 2, f) != 2) {
    fclose(f);
    throw std::runtime_error(std::string("point file too short: ") + path);
  }
  n = head[0];
  d = head[1];
  const size_t cnt = (size_t)n * d, esz = dtype == WANN_DTYPE_F32 ? 4 : 1;
  std::vector<unsigned char> raw(cnt * esz);
  const size_t got = cnt ? fread(raw.data(), esz, cnt, f) : 0;
  fclose(f);
  if (got != cnt) throw std::runtime_error(std::string("point file truncated: ") + path);
  if (dtype == WANN_DTYPE_F32) {
    out.resize(cnt);
    memcpy(out.data(), raw.data(), cnt * 4);
  } else
    out = bytes_to_float(dtype, raw.data(), (int64_t)cnt);
  if (raw_out) raw_out->swap(raw);
}
}  // namespace

struct wann_vamana {
  RawGraph G;
  int dtype = WANN_DTYPE_F32;
  std::mutex mu;
};

wann_vamana *wann_vamana_open(int metric, int dtype, const char *data_path, const char *graph_path, int device) {
  if ((metric != 0 && metric != 1) || dtype < 0 || dtype > 2 || !data_path || !graph_path) {
    fail(WANN_ERR_INVALID, "invalid argument to wann_vamana_open");
    return nullptr;
  }
  if (usable_devices() <= device || device < 0) {
    fail(WANN_ERR_NO_DEVICE, "no usable gfx950 device (this library has no CPU search path)");
    return nullptr;
  }
  try {
    std::unique_ptr<wann_vamana> V(new wann_vamana);
    V->dtype = dtype;
    std::vector<float> pts;
    std::vector<unsigned char> raw;
    int64_t n = 0, d = 0;
    read_point_file(data_path, dtype, pts, n, d, &raw);
    HostGraph g;
    if (!graph_file_load(graph_path, g)) throw std::runtime_error(std::string("cannot read graph file ") + graph_path);
    if (g.n != n) throw std::runtime_error("graph file and point file disagree on the number of points");
    if (g.maxdeg > WANN_MAX_DEGREE) throw std::runtime_error("max_degree > 128 is not supported");
    V->G.load(device, metric, raw.data(), n, d, g.rows.data(), g.maxdeg, 0, n, dtype);
    return V.release();
  } catch (HipError &e) {
    fail(WANN_ERR_HIP, e.what());
  } catch (std::exception &e) {
    fail(WANN_ERR_IO, e.what());
  }
  return nullptr;
}

void wann_vamana_close(wann_vamana *v) { delete v; }
int64_t wann_vamana_num_points(const wann_vamana *v) { return v ? v->G.n : -1; }
int64_t wann_vamana_dim(const wann_vamana *v) { return v ? v->G.d : -1; }

int wann_vamana_batch_search(wann_vamana *V, const void *queries, int64_t nq, int64_t knn, int64_t beam, uint32_t *ids, float *dists) {
  if (!V || nq < 0 || knn <= 0 || beam <= 0 || (nq > 0 && (!queries || !ids || !dists)))
    return fail(WANN_ERR_INVALID, "invalid argument to wann_vamana_batch_search");
  if (beam < knn) return fail(WANN_ERR_INVALID, "beam_width must be at least knn (the reference reads past its beam otherwise)");
  std::lock_guard<std::mutex> lk(V->mu);
  try {
    if (nq == 0) return WANN_OK;
    std::vector<float> qf;
    if (V->dtype != WANN_DTYPE_F32) {
      qf = bytes_to_float(V->dtype, queries, nq * V->G.d);
      queries = qf.data();
    }
    std::vector<int32_t> bid((size_t)nq * beam), bsz((size_t)nq);
    std::vector<float> bd((size_t)nq * beam);
    // QueryParams(knn, beam_width, 1.35, G.size(), G.max_degree()) (vamana_index.cpp:56); query i carries id i (:66)
    V->G.search((const float *)queries, nq, nullptr, beam, V->G.n, V->G.maxdeg, knn, 1.35, bid.data(), bd.data(), bsz.data(), nullptr, nullptr);
    for (int64_t i = 0; i < nq; i++)
      for (int64_t j = 0; j < knn; j++) {
        const bool have = j < bsz[(size_t)i];  // (the reference reads past a shorter beam: defined here as id 2^32-1, FLT_MAX)
        ids[i * knn + j] = have ? (uint32_t)bid[(size_t)(i * beam + j)] : 0xFFFFFFFFu;
        dists[i * knn + j] = have ? bd[(size_t)(i * beam + j)] : 3.402823466e+38f;
      }
  } catch (HipError &e) {
    return fail(WANN_ERR_HIP, e.what());
  } catch (std::exception &e) {
    return fail(WANN_ERR_INVALID, e.what());
  }
  return WANN_OK;
}

int wann_vamana_build_file(int metric, int dtype, const char *data_path, const char *graph_out_path, int64_t max_degree, int64_t limit,
                           double alpha, int device) {
  if ((metric != 0 && metric != 1) || dtype < 0 || dtype > 2 || !data_path || !graph_out_path)
    return fail(WANN_ERR_INVALID, "invalid argument to wann_vamana_build_file");
  try {
    std::vector<float> pts;
    int64_t n = 0, d = 0;
    std::vector<unsigned char> raw;  // (byte point sets are built from their bytes: exact integer distances)
    read_point_file(data_path, dtype, pts, n, d, &raw);
    if (n <= 0 || d <= 0) return fail(WANN_ERR_INVALID, "empty point file");

    // one Vamana graph over the points in file order = the stand-alone post-filter index's graph
    // (knn_index::build_index, vamana/index.h:123-313, BuildParams(R, L, alpha) types.h:94).  The labels only have to be
    // distinct and increasing for the builder to keep file order: float(i) is that below 2^24 points
    if (n > ((int64_t)1 << 24)) return fail(WANN_ERR_UNSUPPORTED, "wann_vamana_build_file: more than 2^24 points are not supported");
    std::vector<float> labels((size_t)n);
    for (int64_t i = 0; i < n; i++) labels[(size_t)i] = (float)i;
    wann_build_params bp{max_degree, limit, alpha, ""};
    wann_index *I = wann_index_create(WANN_KIND_POSTFILTER, metric, dtype, dtype == WANN_DTYPE_F32 ? (const void *)pts.data() : (const void *)raw.data(), n, d,
                                      labels.data(), 1000, 2, 0.5, &bp, device, 0);
    if (!I) return WANN_ERR_HIP;  // (message already set)
    const HostGraph &g = I->H.levels[0][0].g;
    const bool ok = g.n == n && graph_file_save(graph_out_path, g);
    wann_index_destroy(I);
    if (!ok) return fail(WANN_ERR_IO, std::string("cannot write graph file ") + graph_out_path);
  } catch (HipError &e) {
    return fail(WANN_ERR_HIP, e.what());
  } catch (std::exception &e) {
    return fail(WANN_ERR_IO, e.what());
  }
  return WANN_OK;
}

}  // extern "C"
