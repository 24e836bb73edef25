// wann_build_device.h -- argument block of the GPU Vamana build kernels (wann_build_kernels.hip)
// and their launchers, shared with the host orchestration (wann_gpu_build.cpp).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "wann_device.h"

namespace wann {

struct BuildItem {
  int32_t part;   // index into IndexView::parts
  int32_t local;  // local node id inside the partition (or first row of a 64-row tile for k_build_final)
};

struct BuildArgs {
  IndexView ix;
  int32_t *graph_rw;  // the adjacency pool, writable
  const BuildItem *items;
  int32_t nitems;
  int32_t *cursor, *cursor2, *cursor3;
  int32_t L, bits, R;
  double alpha;
  int32_t vis_cap;      // capacity (keys) of the per-wave candidate buffer in LDS
  int32_t *fresh;       // [nitems][R] new out-neighbours of the batch
  int32_t *fresh_cnt;   // [nitems]
  int32_t *g_table;     // per wave slot seen-filter when it does not fit the LDS
  int32_t *err;         // bit 0: a visited list overflowed vis_cap; bit 1: a group overflowed big_cap
  unsigned long long *pair_key, *sorted_key;  // [nitems*R] (partition << 32 | target), ~0 = unused
  int32_t *pair_val, *sorted_val;             // source local id
  int64_t npairs;
  int32_t *gstart, *ngroups;                  // first pair of every (partition, target) group
  int32_t *fallback, *nfallback;              // groups too large for the LDS buffer
  unsigned long long *big_sb;                 // per wave slot global candidate buffer for those
  int64_t big_cap;
  int32_t big;                                // 1: k_build_reverse walks the fallback list with big_sb
  int32_t ref_ties;                           // order exactly equidistant candidates as the reference's std::sort does (wann_stdsort.h)
};

int launch_build_insert(const BuildArgs &a, int blocks, int table_lds, void *stream);
int launch_build_publish(const BuildArgs &a, void *stream);
size_t build_sort_temp_bytes(int64_t npairs);
int launch_build_sort_groups(const BuildArgs &a, void *temp, size_t temp_bytes, void *stream);
int launch_build_reverse(const BuildArgs &a, int blocks, void *stream);
int launch_build_final(const BuildArgs &a, int blocks, void *stream);
int build_lds_bytes_per_wave(int stride, int L, int bits, int table_lds, int vis_cap, int R);
int build_waves_per_block();
const char *build_launch_last_error();

}  // namespace wann
