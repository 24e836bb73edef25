// wann_gemm_device.h -- argument block of the dense prefilter path (wann_gemm_kernels.hip).
#pragma once
#include <stdint.h>

#include "wann_device.h"

namespace wann {

constexpr int kSelect = 32;      // candidates kept per query by MFMA score before the exact re-rank
constexpr int kGemmPointChunk = 2048;  // window positions per tile (multiple of 128)
// (a window hands over three candidates per 64 positions: too short a window could never prove a top 10)
constexpr int kGroupMinQueries = 16, kGroupMinWindow = 1024;

struct GemmGroup {   // queries sharing the window [a, b) of the label argsort
  int64_t a, b;
  int64_t soff;      // offset (floats) of the group's block entries [qcount][steps of 128 positions][2][4] in `scores`
  int32_t qoff;      // the group's query rows are gq[qoff .. qoff + qcount)
  int32_t qcount;
  int32_t tile0;     // the group's tiles are tile0 .. tile0 + nqt * nch - 1: tile = tile0 + ch * nqt + qt
  int32_t nqt, nch;  // 128-query tiles x kGemmPointChunk-position slices
  int32_t pad;
};

// device-side plan of one batch (k_group_*): counts written by the device, read by the kernels that follow
enum { P_NGROUPS = 0, P_NTQ = 1, P_NTILES = 2, P_NSLOTS = 3, P_ANY = 4 /* some slot has enough queries and a long enough window */, P_INTS = 5 };

struct GemmArgs {
  IndexView ix;
  const float *queries;
  const Task *tasks;
  int64_t nq;
  // grouping (open-addressing table over (a, b), `cap` slots, cleared per batch)
  unsigned long long *slot_key;
  int32_t *slot_count, *slot_group;
  int32_t *slot_list;        // the occupied slots, in arrival order
  int32_t cap_mask;
  int32_t *q_slot, *q_rank;  // per query: its slot and its arrival number inside the slot
  int32_t *plan;             // P_*
  unsigned long long *score_used;  // floats of `scores` handed out so far
  GemmGroup *groups;
  int32_t *tile_group;  // per tile: its group (score_cap / 1024 + 1 entries: the smallest group takes 1024 floats of scores per tile)
  int32_t *gq;        // grouped query rows
  int32_t *tq_group;  // per grouped query: its group and its row inside the group
  int32_t *tq_local;
  const float *pnorm2;
  const unsigned int *pnorm2_max_bits;
  float *scores;      // what k_gemm_scores hands to k_rerank's selection: per query, step and half wave the four smallest scores
  int64_t score_cap;  // floats; groups that do not fit any more are left to the exact scan
  int32_t k;
  float acc_factor;  // safety factor on the fp32-accumulation term of the proof's error bound (k_rerank)
  unsigned long long *out_key;
  int32_t *out_cnt;
  int32_t *brute_list, *brute_count;  // exact scan: ungrouped queries + queries whose top-k could not be proven
  unsigned long long *prof;           // dev tool (make PROFILE=1): phase-cycle sums of k_gemm_scores
};

int launch_point_norms(const IndexView &ix, float *norm2, unsigned int *max_bits, void *stream);
int launch_group_windows(const GemmArgs &a, Counters *ctr, void *stream);
int launch_gemm_scores(const GemmArgs &a, int num_cus, void *stream);
int launch_select_rerank(const GemmArgs &a, Counters *ctr, void *stream);
const char *gemm_launch_last_error();

}  // namespace wann
