// wann_gemm_device.h -- argument block of the dense prefilter path (wann_gemm_kernels.hip).
#pragma once
#include <stdint.h>

#include "wann_device.h"

namespace wann {

constexpr int kSelect = 32;  // candidates kept per query by MFMA score before the exact re-rank

struct GemmGroup {   // queries sharing the window [a, b) of the label argsort
  int64_t a, b;
  int64_t soff;      // offset of the group's score matrix [qcount][(b - a) rounded up to 4] in `scores`
  int32_t qoff;      // the group's query rows are gq[qoff .. qoff + qcount)
  int32_t qcount;
};

constexpr int kGemmPointChunk = 2048;  // window positions per workgroup (multiple of 128)

struct GemmTile {
  int32_t group, q0;  // 128-query tile of a group ...
  int64_t p0;         // ... and the first window position of its kGemmPointChunk-point slice
};

struct GemmArgs {
  IndexView ix;
  const float *queries;
  const GemmGroup *groups;
  const GemmTile *tiles;
  int32_t ntiles;
  const int32_t *gq;        // grouped query rows
  const int32_t *tq_group;  // per grouped query: its group and its row inside the group
  const int32_t *tq_local;
  int64_t ntq;
  const float *pnorm2;
  const unsigned int *pnorm2_max_bits;
  float *scores;
  int32_t *sel_pos;   // [ntq][kSelect] window-relative positions
  int32_t *sel_cnt;
  float *sel_cut;     // score of the worst selected candidate (FLT_MAX when the whole window was taken)
  int32_t k;
  unsigned long long *out_key;
  int32_t *out_cnt;
  int32_t *fallback_list, *fallback_count;  // queries whose top-k could not be proven: exact scan
};

int launch_point_norms(const IndexView &ix, float *norm2, unsigned int *max_bits, void *stream);
int launch_gemm_scores(const GemmArgs &a, void *stream);
int launch_select_rerank(const GemmArgs &a, void *stream);
const char *gemm_launch_last_error();

}  // namespace wann
