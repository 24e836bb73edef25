// wann_host_internal.h -- what the host side's translation units share (round 5: wann_host.cpp was one 2 100-line file):
//   wann_host.cpp  device residency of the index, launch geometry, the batch driver (run_batch), the dense prefilter path, GPU build
//   wann_abi.cpp   the C ABI of include/wann.h: index life cycle, blocking / asynchronous / multi-device search calls, RCCL all-gather
//   wann_raw.cpp   one graph over one slice of a point set: wann_raw_beam_search and the unfiltered VamanaIndex API
// Internal: nothing here is part of the boundary (include/wann.h is).
#pragma once
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

namespace wann_host {

extern thread_local std::string g_err;
int fail(int code, const std::string &msg);
int usable_devices();
int hash_bits(int64_t beam);  // beamSearch.h:66
void convert_rows(const HostGraph &g, int rs, int32_t *out);  // reference in-memory rows -> device rows: rs ints per row, neighbours packed, -1 padded

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
  DevBuf<int64_t> vroute;            // QueryParams::verbose on a tree class: the descent's dump (RouteArgs::vroute)
  DevBuf<int32_t> gat_send, gat_recv;  // wann_batch_search_allgather: this replica's [2][cap][k] planes / everybody's [world][2][cap][k]
  int32_t big_stride = 0;
  int32_t *h_ints = nullptr;  // pinned
  Counters *h_ctr = nullptr;  // pinned
  std::vector<hipEvent_t> ev;
  hipEvent_t ev_side = nullptr;   // end of the companion (big) launch on the index's side stream
  hipEvent_t ev_route = nullptr;  // list sizes of k_route are on the host
  // fenwick / three_split: the end scans (k_brute) run BESIDE the graph searches of the same batch on a stream of the lane's own
  hipStream_t scan_stream = nullptr;
  hipEvent_t ev_scan = nullptr;
  ~Workspace() {
    if (h_ints) (void)hipHostFree(h_ints);
    if (h_ctr) (void)hipHostFree(h_ctr);
    for (auto e : ev) (void)hipEventDestroy(e);
    if (ev_side) (void)hipEventDestroy(ev_side);
    if (ev_route) (void)hipEventDestroy(ev_route);
    if (ev_scan) (void)hipEventDestroy(ev_scan);
    if (scan_stream) (void)hipStreamDestroy(scan_stream);
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


}  // namespace wann_host

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
  wann_host::Workspace ws;
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
  wann_index();   // (both out of line, wann_abi.cpp: AsyncLane / Rccl are complete types only there)
  ~wann_index();
};

namespace wann_host {

Tuning snapshot_tuning(wann_index &I);  // the index's switches for one call (WANN_TEST_HOOKS=1: re-read from the environment first)
void upload_index(wann_index &I);
int ensure_filter_scratch(DevBuf<int32_t> &table, DevBuf<int32_t> &epoch, DevBuf<uint32_t> &seen, int64_t &layout, int slots,
                          int table_bits, int64_t seen_words, hipStream_t st, bool any_slots = false);
struct RoundCfg {
  LaunchCfg lc;
  bool big_lds;      // the one-wave-per-workgroup kernel
  int slots;
  int pool_bytes;    // per-wave LDS pool of this launch
  int table_bits;    // per-slot global seen-filter of 4 << table_bits bytes (0 = none needed)
  int64_t beam_cap;  // per-slot global beam entries (0 = none needed)
};
RoundCfg config_for(const wann_index &I, const Tuning &T, int64_t first_beam, int64_t cap, int64_t work_items, bool big_lds = false, bool force_table = false,
                    bool legacy = false, int base_pool = kSearchPoolBytes);
int lean_pool_bytes(const wann_index &I, const Tuning &T);
int method_code(const char *m);
// W / side / last: the lane of this batch (the index's own members for the blocking calls, an AsyncLane's for the asynchronous one)
void run_batch(wann_index &I, Workspace &W, hipStream_t side, wann_counters &last, const float *d_queries, const float *d_ranges, int64_t nq,
               int64_t qid_base, const char *method, const wann_query_params &qp, uint32_t *d_ids, float *d_dists, hipStream_t st, const Tuning &T,
               const int64_t *d_qids = nullptr);
void build_pending(wann_index &I, std::vector<HostPart *> &pending);
std::vector<float> bytes_to_float(int dtype, const void *src, int64_t count);
BuildSpec make_spec(int kind, int metric, int dtype, int64_t n, int64_t d, int32_t cutoff, double split_factor, double shift_factor,
                    const wann_build_params *bp, int threads);

}  // namespace wann_host
using namespace wann_host;

