// wann_build_kernels.hip -- gfx950 kernels of the GPU Vamana build (reference algorithm:
// ParlayANN/algorithms/vamana/index.h:61-135,211-313).  The build runs in lock-step rounds over
// ALL partitions of an index (round r = batch r of every unfinished partition, same batch
// schedule as a stand-alone build), directly on the adjacency pool that the search kernels use:
//
//   k_build_insert   per inserted point: beam search on the snapshot (beam L, visited list kept)
//                    + robustPrune(visited U current out-neighbours)            index.h:268-274, 61-108
//   k_build_publish  write the batch's new out-neighbour rows                    index.h:287-289
//   k_build_pairs    reverse edges (target <- source) as sortable keys           index.h:277-286
//   (hipcub radix sort: stable, so sources stay in batch order inside a target)  index.h:290
//   k_build_groups   one group per (partition, target)
//   k_build_reverse  append the group's sources when the row has room, else robustPrune   index.h:297-306
//   k_build_final    per-node neighbour sort by distance to the node             index.h:131-134
//
// Every tie breaks by (distance, id) and every distance uses the reference-order routines of
// wann_wave.h, so the graphs are byte-identical to the host builder's (wann_build.cpp) -- which
// the tests check -- for any device occupancy.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>

#include "wann_build_device.h"
#include "wann_wave.h"

namespace wann {

constexpr int kBuildWaves = 2;  // waves per workgroup of the build kernels

__device__ __forceinline__ u64 mk_key(float dist, int id) { return ((u64)fkey(dist) << 32) | ((u64)(uint32_t)id << 1); }
__device__ __forceinline__ int key_id(u64 k) { return (int)((uint32_t)k >> 1); }
__device__ __forceinline__ float key_dist(u64 k) { return funkey((uint32_t)(k >> 32)); }

// robustPrune over the unsorted candidate keys sb[0..nc) (bit 0 = consumed / pruned).  Selecting
// "the next candidate in (dist,id) order" is an arg-min over the live keys, which is what walking
// the sorted list does in the reference; the alpha test kills live candidates c with
// alpha * d(p*, c) <= d(p, c) (index.h:93-103).  Returns the number of selected neighbours (out[]).
template <int METRIC>
__device__ __forceinline__ int wave_prune(const IndexView &ix, int64_t row_off, int p, u64 *sb, int nc, int R,
                                          double alpha, const WaveLds &L, int32_t *out) {
  const int lane = lane_id();
  int outc = 0;
  for (;;) {
    // arg-min over live keys, ties to the lowest position
    u64 best = ~0ull;
    int bpos = 0x7fffffff;
    for (int x = lane; x < nc; x += 64) {
      u64 k = sb[x];
      if (!(k & 1ull) && (k < best)) {
        best = k;
        bpos = x;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      u64 ok = ((u64)(uint32_t)__shfl_xor((int)(uint32_t)(best >> 32), o) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)best, o);
      int op = __shfl_xor(bpos, o);
      if (ok < best || (ok == best && op < bpos)) {
        best = ok;
        bpos = op;
      }
    }
    if (bpos == 0x7fffffff) break;  // nothing alive
    const int ps = key_id(best);
    if (lane == 0) sb[bpos] = best | 1ull;  // consumed
    WAVE_SYNC();
    if (ps == p) continue;
    if (lane == 0) out[outc] = ps;
    outc++;
    if (outc >= R) break;
    for (int i = lane; i < ix.stride; i += 64) L.qv[i] = ix.points[(row_off + ps) * (int64_t)ix.stride + i];
    WAVE_SYNC();
    for (int base = 0; base < nc; base += 64) {
      const int x = base + lane;
      u64 kx = (x < nc) ? sb[x] : 1ull;
      const bool al = !(kx & 1ull);
      const u64 am = ballot64(al);
      if (!am) continue;
      const int cnt = popc64(am);
      const int slot = popc64(am & lanemask_lt());
      if (al) {
        L.cand_id[slot] = key_id(kx);
        L.cand_key[slot] = (u64)x;
      }
      WAVE_SYNC();
      const float dsp = wave_distances<METRIC>(ix, L.cand_id, L.cand_dist, L.qv, cnt, row_off);
      if (lane < cnt) {
        const int xx = (int)L.cand_key[lane];
        const u64 kk = sb[xx];
        if (alpha * (double)dsp <= (double)key_dist(kk)) sb[xx] = kk | 1ull;
      }
      WAVE_SYNC();
    }
  }
  WAVE_SYNC();
  return outc;
}

__device__ __forceinline__ int build_lds_per_wave(int stride, int L, int bits, bool table_lds, int vis_cap, int R) {
  int bytes = wave_lds_common_bytes(stride);
  bytes += ((L + 1) & ~1) * 8;
  if (table_lds) bytes += 4 << bits;
  bytes += vis_cap * 8;
  bytes += ((R + 3) & ~3) * 4;
  return (bytes + 15) & ~15;
}

// ------------------------------------------------------------------------------------------------
template <int METRIC, bool TABLE_LDS>
__global__ __launch_bounds__(64 * kBuildWaves) void k_build_insert(BuildArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const IndexView &ix = A.ix;
  const int lane = lane_id();
  const int wib = threadIdx.x >> 6;
  const int slot = blockIdx.x * kBuildWaves + wib;
  const int per_wave = build_lds_per_wave(ix.stride, A.L, A.bits, TABLE_LDS, A.vis_cap, A.R);
  unsigned char *base = smem + (size_t)wib * per_wave;
  const WaveLds L = carve_wave_lds(base, ix.stride, A.L, true);
  unsigned char *after = reinterpret_cast<unsigned char *>(L.ltable) + (TABLE_LDS ? (4 << A.bits) : 0);
  u64 *sb = reinterpret_cast<u64 *>(after);
  int32_t *out = reinterpret_cast<int32_t *>(after + (size_t)A.vis_cap * 8);
  int32_t *gtable = TABLE_LDS ? nullptr : A.g_table + ((size_t)slot << A.bits);

  for (;;) {
    const int t = wave_ticket(A.cursor);
    if (t >= A.nitems) break;
    const BuildItem item = A.items[t];
    const PartDesc part = ix.parts[item.part];
    const int p = item.local;
    const int64_t row_off = part.start;
    for (int i = lane; i < ix.stride; i += 64) L.qv[i] = ix.points[(row_off + p) * (int64_t)ix.stride + i];
    WAVE_SYNC();
    int m;
    long long nvis, ncmp;
    // builder-side quirk: the searched point's own id is its PARENT index (App. B #17 of SURVEY.md)
    wave_beam_search<METRIC, TABLE_LDS, true, true>(ix, part, L, nullptr, gtable, A.L, A.bits, row_off + p,
                                                    (int64_t)part.n, (int)A.R, sb, A.vis_cap - 64, m, nvis, ncmp);
    if (nvis > A.vis_cap - 64) {  // visited list does not fit: the host rebuilds this partition
      if (lane == 0) atomicOr(A.err, 1);
      if (lane == 0) A.fresh_cnt[t] = 0;
      continue;
    }
    int nc = (int)nvis;
    // candidates += current out-neighbours of p with their distances (robustPrune add = true)
    int a = -1;
    if (lane < ix.rs) a = ix.graph[(part.row_base + p) * (int64_t)ix.rs + lane];
    const u64 vm = ballot64(a >= 0);
    const int deg = popc64(vm);
    if (a >= 0) L.cand_id[popc64(vm & lanemask_lt())] = a;
    WAVE_SYNC();
    const float dn = wave_distances<METRIC>(ix, L.cand_id, L.cand_dist, L.qv, deg, row_off);
    if (lane < deg) sb[nc + lane] = mk_key(dn, L.cand_id[lane]);
    nc += deg;
    WAVE_SYNC();
    const int outc = wave_prune<METRIC>(ix, row_off, p, sb, nc, A.R, A.alpha, L, out);
    if (lane < outc) A.fresh[(size_t)t * A.R + lane] = out[lane];
    if (lane == 0) A.fresh_cnt[t] = outc;
    WAVE_SYNC();
  }
}

// one thread per (item, slot): publish rows, emit reverse-edge pairs
__global__ void k_build_publish(BuildArgs A) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int rs = A.ix.rs;
  if (g >= (int64_t)A.nitems * rs) return;
  const int t = (int)(g / rs), j = (int)(g % rs);
  const BuildItem item = A.items[t];
  const PartDesc part = A.ix.parts[item.part];
  const int cnt = A.fresh_cnt[t];
  const int v = (j < cnt) ? A.fresh[(size_t)t * A.R + j] : -1;
  A.graph_rw[(part.row_base + item.local) * (int64_t)rs + j] = v;
  if (j < A.R) {
    const size_t pi = (size_t)t * A.R + j;
    A.pair_key[pi] = (j < cnt) ? (((u64)(uint32_t)item.part << 32) | (uint32_t)v) : ~0ull;
    A.pair_val[pi] = item.local;
  }
}

__global__ void k_build_groups(const u64 *keys, int64_t npairs, int32_t *gstart, int32_t *ngroups) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npairs) return;
  const u64 k = keys[i];
  if (k == ~0ull) return;
  if (i == 0 || keys[i - 1] != k) gstart[atomicAdd(ngroups, 1)] = (int32_t)i;
}

template <int METRIC>
__global__ __launch_bounds__(64 * kBuildWaves) void k_build_reverse(BuildArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const IndexView &ix = A.ix;
  const int lane = lane_id();
  const int wib = threadIdx.x >> 6;
  const int per_wave = build_lds_per_wave(ix.stride, 0, 0, false, A.vis_cap, A.R);
  unsigned char *base = smem + (size_t)wib * per_wave;
  const WaveLds L = carve_wave_lds(base, ix.stride, 0, true);
  u64 *lsb = reinterpret_cast<u64 *>(L.ltable);
  int32_t *out = reinterpret_cast<int32_t *>(reinterpret_cast<unsigned char *>(lsb) + (size_t)A.vis_cap * 8);
  const int slot = blockIdx.x * kBuildWaves + wib;
  // second pass (A.big): the groups that did not fit the LDS buffer, candidates in global scratch
  u64 *sb = A.big ? (A.big_sb + (size_t)slot * A.big_cap) : lsb;
  const int64_t cap = A.big ? A.big_cap : (int64_t)A.vis_cap;
  const int ngroups = A.big ? *A.nfallback : *A.ngroups;
  const int rs = ix.rs;

  for (;;) {
    const int gi = wave_ticket(A.big ? A.cursor3 : A.cursor2);
    if (gi >= ngroups) break;
    const int g = A.big ? A.fallback[gi] : gi;
    const int b = A.gstart[g];
    const u64 key = A.sorted_key[b];
    const int pidx = (int)(key >> 32), tgt = (int)(uint32_t)key;
    int64_t lo = b, hi = A.npairs;  // first index past the run of `key` (keys are sorted)
    while (lo < hi) {
      int64_t mid = (lo + hi) >> 1;
      if (A.sorted_key[mid] <= key) lo = mid + 1;
      else hi = mid;
    }
    const int gsize = (int)(lo - b);
    const PartDesc part = ix.parts[pidx];
    const int64_t row_off = part.start;
    int32_t *row = A.graph_rw + (part.row_base + tgt) * (int64_t)rs;
    int a = -1;
    if (lane < rs) a = row[lane];
    const u64 vm = ballot64(a >= 0);
    const int deg = popc64(vm);
    if (deg + gsize <= A.R) {  // room: append in batch order (index.h:299-301)
      for (int j = lane; j < gsize; j += 64) row[deg + j] = A.sorted_val[b + j];
      continue;
    }
    if ((int64_t)deg + gsize > cap) {  // hub node with a huge group: second pass with a global buffer
      if (lane == 0) {
        if (A.big) atomicOr(A.err, 2);
        else A.fallback[atomicAdd(A.nfallback, 1)] = g;
      }
      continue;
    }
    // robustPrune(tgt, sources U current row) (index.h:302-305)
    for (int i = lane; i < ix.stride; i += 64) L.qv[i] = ix.points[(row_off + tgt) * (int64_t)ix.stride + i];
    if (a >= 0) L.cand_id[popc64(vm & lanemask_lt())] = a;
    WAVE_SYNC();
    const float dr = wave_distances<METRIC>(ix, L.cand_id, L.cand_dist, L.qv, deg, row_off);
    if (lane < deg) sb[gsize + lane] = mk_key(dr, L.cand_id[lane]);
    WAVE_SYNC();
    for (int c0 = 0; c0 < gsize; c0 += 64) {
      const int cnt = (gsize - c0) < 64 ? (gsize - c0) : 64;
      int sid = 0;
      if (lane < cnt) sid = A.sorted_val[b + c0 + lane];
      L.cand_id[lane] = sid;
      WAVE_SYNC();
      const float ds = wave_distances<METRIC>(ix, L.cand_id, L.cand_dist, L.qv, cnt, row_off);
      if (lane < cnt) sb[c0 + lane] = mk_key(ds, sid);
      WAVE_SYNC();
    }
    const int outc = wave_prune<METRIC>(ix, row_off, tgt, sb, gsize + deg, A.R, A.alpha, L, out);
    if (lane < rs) row[lane] = (lane < outc) ? out[lane] : -1;
    WAVE_SYNC();
  }
}

// final neighbour sort: one wave per pool row
template <int METRIC>
__global__ __launch_bounds__(64 * kBuildWaves) void k_build_final(BuildArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const IndexView &ix = A.ix;
  const int lane = lane_id();
  const int wib = threadIdx.x >> 6;
  const int per_wave = build_lds_per_wave(ix.stride, 0, 0, false, 0, 0);
  const WaveLds L = carve_wave_lds(smem + (size_t)wib * per_wave, ix.stride, 0, true);
  const int rs = ix.rs;
  for (;;) {
    const int t = wave_ticket(A.cursor);
    if (t >= A.nitems) break;
    const BuildItem item = A.items[t];  // here: (partition, first local row) of a 64-row tile
    const PartDesc part = ix.parts[item.part];
    const int64_t row_off = part.start;
    const int endl = (item.local + 64) < part.n ? (item.local + 64) : part.n;
    for (int node = item.local; node < endl; node++) {
      int32_t *row = A.graph_rw + (part.row_base + node) * (int64_t)rs;
      int a = -1;
      if (lane < rs) a = row[lane];
      const u64 vm = ballot64(a >= 0);
      const int deg = popc64(vm);
      if (deg == 0) continue;
      for (int i = lane; i < ix.stride; i += 64) L.qv[i] = ix.points[(row_off + node) * (int64_t)ix.stride + i];
      if (a >= 0) L.cand_id[popc64(vm & lanemask_lt())] = a;
      WAVE_SYNC();
      const float dn = wave_distances<METRIC>(ix, L.cand_id, L.cand_dist, L.qv, deg, row_off);
      const int nid = (lane < deg) ? L.cand_id[lane] : 0;
      const u64 key = (lane < deg) ? mk_key(dn, nid) : ~0ull;
      int rank = 0;
      for (int l = 0; l < deg; l++) {
        const u64 kl = rdlane64(key, l);
        rank += (kl < key || (kl == key && l < lane)) ? 1 : 0;
      }
      WAVE_SYNC();
      if (lane < deg) row[rank] = nid;
      WAVE_SYNC();
    }
  }
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
static thread_local const char *g_berr = "";
const char *build_launch_last_error() { return g_berr; }
static int bcheck(hipError_t e) {
  if (e != hipSuccess) {
    g_berr = hipGetErrorString(e);
    return 1;
  }
  return 0;
}

int build_lds_bytes_per_wave(int stride, int L, int bits, int table_lds, int vis_cap, int R) {
  int bytes = ((stride * 4 + 15) & ~15) + 64 * 8 + 64 * 4 + 64 * 4;
  bytes += ((L + 1) & ~1) * 8;
  if (table_lds) bytes += 4 << bits;
  bytes += vis_cap * 8;
  bytes += ((R + 3) & ~3) * 4;
  return (bytes + 15) & ~15;
}
int build_waves_per_block() { return kBuildWaves; }

template <typename K>
static int set_lds(K kern, size_t lds) {
  if (lds > 48 * 1024)
    return bcheck(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  return 0;
}

int launch_build_insert(const BuildArgs &a, int blocks, int table_lds, void *stream) {
  if (blocks <= 0 || a.nitems <= 0) return 0;
  size_t lds = (size_t)build_lds_bytes_per_wave(a.ix.stride, a.L, a.bits, table_lds, a.vis_cap, a.R) * kBuildWaves;
  dim3 grid(blocks), block(64 * kBuildWaves);
  hipStream_t s = (hipStream_t)stream;
#define WANN_BL(M, T)                              \
  do {                                             \
    auto kern = k_build_insert<M, T>;              \
    if (set_lds(kern, lds)) return 1;              \
    hipLaunchKernelGGL(kern, grid, block, lds, s, a); \
  } while (0)
  if (a.ix.metric == 1) {
    if (table_lds) WANN_BL(1, true);
    else WANN_BL(1, false);
  } else {
    if (table_lds) WANN_BL(0, true);
    else WANN_BL(0, false);
  }
#undef WANN_BL
  return bcheck(hipGetLastError());
}

int launch_build_publish(const BuildArgs &a, void *stream) {
  if (a.nitems <= 0) return 0;
  int64_t n = (int64_t)a.nitems * a.ix.rs;
  hipLaunchKernelGGL(k_build_publish, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  return bcheck(hipGetLastError());
}

size_t build_sort_temp_bytes(int64_t npairs) {
  size_t bytes = 0;
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const u64 *)nullptr, (u64 *)nullptr, (const int32_t *)nullptr,
                                     (int32_t *)nullptr, (int)npairs);
  return bytes;
}

int launch_build_sort_groups(const BuildArgs &a, void *temp, size_t temp_bytes, void *stream) {
  if (a.npairs <= 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  if (bcheck(hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, (const u64 *)a.pair_key, a.sorted_key,
                                                (const int32_t *)a.pair_val, a.sorted_val, (int)a.npairs, 0, 64, s)))
    return 1;
  hipLaunchKernelGGL(k_build_groups, dim3((unsigned)((a.npairs + 255) / 256)), dim3(256), 0, s, (const u64 *)a.sorted_key,
                     a.npairs, a.gstart, a.ngroups);
  return bcheck(hipGetLastError());
}

int launch_build_reverse(const BuildArgs &a, int blocks, void *stream) {
  if (blocks <= 0) return 0;
  size_t lds = (size_t)build_lds_bytes_per_wave(a.ix.stride, 0, 0, 0, a.vis_cap, a.R) * kBuildWaves;
  dim3 grid(blocks), block(64 * kBuildWaves);
  hipStream_t s = (hipStream_t)stream;
  if (a.ix.metric == 1) {
    auto kern = k_build_reverse<1>;
    if (set_lds(kern, lds)) return 1;
    hipLaunchKernelGGL(kern, grid, block, lds, s, a);
  } else {
    auto kern = k_build_reverse<0>;
    if (set_lds(kern, lds)) return 1;
    hipLaunchKernelGGL(kern, grid, block, lds, s, a);
  }
  return bcheck(hipGetLastError());
}

int launch_build_final(const BuildArgs &a, int blocks, void *stream) {
  if (blocks <= 0 || a.nitems <= 0) return 0;
  size_t lds = (size_t)build_lds_bytes_per_wave(a.ix.stride, 0, 0, 0, 0, 0) * kBuildWaves;
  dim3 grid(blocks), block(64 * kBuildWaves);
  if (a.ix.metric == 1) hipLaunchKernelGGL(k_build_final<1>, grid, block, lds, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(k_build_final<0>, grid, block, lds, (hipStream_t)stream, a);
  return bcheck(hipGetLastError());
}

}  // namespace wann
