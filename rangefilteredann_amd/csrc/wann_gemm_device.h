// wann_gemm_device.h -- argument block of the dense prefilter path (wann_gemm_kernels.hip).
#pragma once
#include <stdint.h>

#include "wann_device.h"

namespace wann {

constexpr int kSelect = 32;      // candidates kept per query by MFMA score before the exact re-rank
constexpr int kCandCap = 64;     // candidate slots per (query, window slice) inside a tile and in the hand-over to k_rerank
constexpr int kMaxChunks = 8;    // window slices per group (the slice length grows beyond 8 x kGemmPointChunk points)
constexpr int kGemmPointChunk = 2048;  // window positions per workgroup (multiple of 128)
constexpr int kGroupMinQueries = 16, kGroupMinWindow = 64;

struct GemmGroup {   // queries sharing the window [a, b) of the label argsort
  int64_t a, b;
  int32_t qoff;      // the group's query rows are gq[qoff .. qoff + qcount)
  int32_t qcount;
  int32_t nch;       // number of window slices (<= kMaxChunks) ...
  int32_t chunk;     // ... of this many positions each (multiple of 128)
};

struct GemmTile {
  int32_t group, q0;  // 128-query tile of a group ...
  int32_t ch;         // ... and the window slice it scores
};

// device-side plan of one batch (k_group_*): counts written by the device, read by the kernels that follow
enum { P_NGROUPS = 0, P_NTQ = 1, P_NTILES = 2, P_INTS = 4 };

struct GemmArgs {
  IndexView ix;
  const float *queries;
  const Task *tasks;
  int64_t nq;
  // grouping (open-addressing table over (a, b), `cap` slots, cleared per batch)
  unsigned long long *slot_key;
  int32_t *slot_count, *slot_group;
  int32_t cap_mask;
  int32_t *q_slot, *q_rank;  // per query: its slot and its arrival number inside the slot
  int32_t *plan;             // P_*
  GemmGroup *groups;
  GemmTile *tiles;
  int32_t *gq;        // grouped query rows
  int32_t *tq_group;  // per grouped query: its group and its row inside the group
  int32_t *tq_local;
  const float *pnorm2;
  const unsigned int *pnorm2_max_bits;
  // hand-over from k_gemm_select to k_rerank, per (grouped query, slice): unsorted candidate keys (score, position)
  unsigned long long *cand_key;  // [ntq][kMaxChunks][kCandCap]
  int32_t *cand_cnt;             // [ntq][kMaxChunks]
  float *cand_cut;               // [ntq][kMaxChunks]: every position of the slice that is not in the list scores >= this
  unsigned int *thr_shared;      // [ntq]: order-preserving bits of the lowest cut any slice of the query reached so far
  int32_t k;
  unsigned long long *out_key;
  int32_t *out_cnt;
  int32_t *brute_list, *brute_count;  // exact scan: ungrouped queries + queries whose top-k could not be proven
};

int launch_point_norms(const IndexView &ix, float *norm2, unsigned int *max_bits, void *stream);
int launch_group_windows(const GemmArgs &a, Counters *ctr, void *stream);
int launch_gemm_select(const GemmArgs &a, int num_cus, void *stream);
int launch_rerank(const GemmArgs &a, Counters *ctr, void *stream);
const char *gemm_launch_last_error();

}  // namespace wann
