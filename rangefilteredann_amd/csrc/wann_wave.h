// wann_wave.h -- wave64 device building blocks shared by the search kernels (wann_kernels.hip) and
// the graph-build kernels (wann_build_kernels.hip): lane utilities, the reference-order distance
// routines, the sorted-list merge and the beam-search core (one wavefront = one search).
// Reference semantics: ParlayANN/algorithms/utils/beamSearch.h:51-184.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "wann_device.h"

namespace wann {

typedef unsigned long long u64;

// Per-phase cycle accounting of the search cores is a compile-time option (make PROFILE=1, dev tool
// tools/phase_profile.py): its accumulators cost the production kernels a dozen scalar registers.
#ifdef WANN_PHASE_PROFILE
#define WANN_PROF_PTR(p) (p)
#else
#define WANN_PROF_PTR(p) ((unsigned long long *)nullptr)
#endif

#define WANN_LIKELY(x) __builtin_expect(!!(x), 1)
#define WANN_UNLIKELY(x) __builtin_expect(!!(x), 0)

#define WAVE_SYNC()                                            \
  do {                                                         \
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");     \
    __builtin_amdgcn_wave_barrier();                           \
  } while (0)

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ u64 ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int rdlane(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ u64 rdlane64(u64 v, int l) {
  uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
  uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l);
  return ((u64)hi << 32) | lo;
}
// a wave-uniform 64-bit value the compiler can not prove uniform (it came from a vector or atomic load): into scalar registers
__device__ __forceinline__ long long uni64(long long v) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((u64)v >> 32));
  return (long long)(((u64)hi << 32) | lo);
}
template <typename T>
__device__ __forceinline__ T *uniptr(T *p) {
  return reinterpret_cast<T *>(uni64((long long)reinterpret_cast<uintptr_t>(p)));
}
__device__ __forceinline__ int ctz64(u64 m) { return __builtin_ctzll(m); }
__device__ __forceinline__ int popc64(u64 m) { return __builtin_popcountll(m); }
__device__ __forceinline__ u64 lanemask_lt() { return ((u64)1 << lane_id()) - 1; }

// One work-list ticket per wave.  The lane election must not look loop invariant to the compiler:
// hipcc (ROCm 7.2) otherwise unswitches the persistent loop on `lane == 0` and the non-zero lanes
// spin on a stale ticket (readfirstlane then runs under a partial exec mask).  The volatile asm
// keeps the predicate inside the loop.
__device__ __forceinline__ int wave_ticket(int32_t *cursor) {
  int lane;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
  int t = 0;
  if (lane == 0) t = atomicAdd(cursor, 1);
  return __builtin_amdgcn_readfirstlane(t);
}

// order preserving float -> uint32 (ascending)
__device__ __forceinline__ uint32_t fkey(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float funkey(uint32_t k) {
  return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k);
}

// parlay::hash64_2 (parlay/utilities.h:145-150)
__device__ __forceinline__ u64 hash64_2(u64 x) {
  x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
  x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
  return x ^ (x >> 31);
}

// --------------------------------------------------------------------------------------------
// distances in the reference's evaluation order
// --------------------------------------------------------------------------------------------
// Squared L2 (NSGDist.h:33-69): 8 accumulators (the AVX lanes); a lane PAIR owns one candidate:
// lane h = lane&1 carries accumulators 4h..4h+3 and walks the 8-float blocks in the reference's
// order (odd block count: last block first).  Returns the full distance in the odd lane.
template <int NB>
__device__ __forceinline__ float l2_pair(const float *__restrict__ prow, const float *qv, int D8,
                                         int h, bool active) {
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (active) {
    const bool odd = D8 & 1;
    for (int i0 = 0; i0 < D8; i0 += NB) {
      float4 buf[NB];
#pragma unroll
      for (int j = 0; j < NB; j++) {
        int i = i0 + j;
        if (i < D8) {
          int b = odd ? (i == 0 ? D8 - 1 : i - 1) : i;
          buf[j] = *reinterpret_cast<const float4 *>(prow + 8 * b + 4 * h);
        }
      }
#pragma unroll
      for (int j = 0; j < NB; j++) {
        int i = i0 + j;
        if (i < D8) {
          int b = odd ? (i == 0 ? D8 - 1 : i - 1) : i;
          float4 q = *reinterpret_cast<const float4 *>(qv + 8 * b + 4 * h);
          float t;
          t = buf[j].x - q.x; a0 = fmaf(t, t, a0);
          t = buf[j].y - q.y; a1 = fmaf(t, t, a1);
          t = buf[j].z - q.z; a2 = fmaf(t, t, a2);
          t = buf[j].w - q.w; a3 = fmaf(t, t, a3);
        }
      }
    }
  }
  float s = ((a0 + a1) + a2) + a3;          // even lane: ((l0+l1)+l2)+l3
  float other = __shfl_xor(s, 1);           // odd lane receives the even lane's partial
  return (((other + a0) + a1) + a2) + a3;   // odd lane: ((((s+l4)+l5)+l6)+l7)
}

// Same arithmetic with the block count known at compile time: no branches, the row's loads, then the
// query's LDS reads, are issued back to back (the hot path for d = 128 / 96 / 100 / 64 / 32).
typedef float f32x2 __attribute__((ext_vector_type(2)));

// acc += (p - q)^2 element-wise with one rounding per element (fma): packed fp32 instructions
// (v_pk_add_f32 / v_pk_fma_f32) compute exactly what two scalar ops compute.
__device__ __forceinline__ void sq_acc(f32x2 &lo, f32x2 &hi, const float4 &p, const float4 &q) {
  f32x2 t0 = f32x2{p.x, p.y} - f32x2{q.x, q.y};
  f32x2 t1 = f32x2{p.z, p.w} - f32x2{q.z, q.w};
  lo = __builtin_elementwise_fma(t0, t0, lo);
  hi = __builtin_elementwise_fma(t1, t1, hi);
}

template <int D8C>
__device__ __forceinline__ float l2_pair_ct(const float *__restrict__ prow, const float *qv, int h, bool active) {
  // inactive lanes are handed a valid row (node 0 of the partition) and compute a value nobody
  // reads: no per-load exec-mask branches
  (void)active;
  constexpr bool odd = D8C & 1;
  f32x2 alo = {0.f, 0.f}, ahi = {0.f, 0.f};
  float4 buf[D8C];
#pragma unroll
  for (int i = 0; i < D8C; i++) {
    const int b = odd ? (i == 0 ? D8C - 1 : i - 1) : i;
    buf[i] = *reinterpret_cast<const float4 *>(prow + 8 * b + 4 * h);
  }
#pragma unroll
  for (int i = 0; i < D8C; i++) {
    const int b = odd ? (i == 0 ? D8C - 1 : i - 1) : i;
    const float4 q = *reinterpret_cast<const float4 *>(qv + 8 * b + 4 * h);
    sq_acc(alo, ahi, buf[i], q);
  }
  float s = ((alo.x + alo.y) + ahi.x) + ahi.y;
  float other = __shfl_xor(s, 1);
  return (((other + alo.x) + alo.y) + ahi.x) + ahi.y;
}

// Two candidates per lane pair (s and s + 32) with every block of BOTH rows in flight before the
// first use: one HBM round trip scores up to 64 candidates.
template <int D8C>
__device__ __forceinline__ void l2_pair2_ct(const float *__restrict__ prow0, const float *__restrict__ prow1,
                                            const float *qv, int h, float &d0, float &d1) {
  constexpr bool odd = D8C & 1;
  float4 b0[D8C], b1[D8C];
#pragma unroll
  for (int i = 0; i < D8C; i++) {
    const int b = odd ? (i == 0 ? D8C - 1 : i - 1) : i;
    b0[i] = *reinterpret_cast<const float4 *>(prow0 + 8 * b + 4 * h);
  }
#pragma unroll
  for (int i = 0; i < D8C; i++) {
    const int b = odd ? (i == 0 ? D8C - 1 : i - 1) : i;
    b1[i] = *reinterpret_cast<const float4 *>(prow1 + 8 * b + 4 * h);
  }
  f32x2 alo = {0.f, 0.f}, ahi = {0.f, 0.f}, clo = {0.f, 0.f}, chi = {0.f, 0.f};
#pragma unroll
  for (int i = 0; i < D8C; i++) {
    const int b = odd ? (i == 0 ? D8C - 1 : i - 1) : i;
    const float4 q = *reinterpret_cast<const float4 *>(qv + 8 * b + 4 * h);
    sq_acc(alo, ahi, b0[i], q);
    sq_acc(clo, chi, b1[i], q);
  }
  float s = ((alo.x + alo.y) + ahi.x) + ahi.y;
  float other = __shfl_xor(s, 1);
  d0 = (((other + alo.x) + alo.y) + ahi.x) + ahi.y;
  s = ((clo.x + clo.y) + chi.x) + chi.y;
  other = __shfl_xor(s, 1);
  d1 = (((other + clo.x) + clo.y) + chi.x) + chi.y;
}

// Negative inner product (mips_point.h:60-66 as compiled): ONE running scalar per candidate, products rounded
// then added in index order for the first 8*floor(d/8) elements, fused multiply-adds for the tail.
// A lane PAIR owns a candidate (like the L2 routine): lane h loads the float4s 2t+h of the row -- 32
// contiguous bytes per pair and load instruction instead of 16 bytes from 64 different rows -- and multiplies
// them by its part of the query; the chain of additions is the same in both lanes, each addend fetched from
// the lane that holds it by a DPP quad permutation (no extra instruction, no LDS).  The order of the
// additions is exactly the sequential one.
template <bool ODD>
__device__ __forceinline__ float pair_lane(float v) {  // the value lane 2c+ODD of my pair holds
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ODD ? 0xF5 : 0xA0, 0xF, 0xF, true));
}

// one pair step: float4 2t (even lane) and 2t+1 (odd lane) of row and query; `tail` = fused part (wave-uniform)
__device__ __forceinline__ float mips_step(float r, const float4 &p, const float4 &q, bool tail) {
  if (!tail) {
    const float px = __fmul_rn(q.x, p.x), py = __fmul_rn(q.y, p.y), pz = __fmul_rn(q.z, p.z), pw = __fmul_rn(q.w, p.w);
    r = __fadd_rn(r, pair_lane<false>(px));
    r = __fadd_rn(r, pair_lane<false>(py));
    r = __fadd_rn(r, pair_lane<false>(pz));
    r = __fadd_rn(r, pair_lane<false>(pw));
    r = __fadd_rn(r, pair_lane<true>(px));
    r = __fadd_rn(r, pair_lane<true>(py));
    r = __fadd_rn(r, pair_lane<true>(pz));
    r = __fadd_rn(r, pair_lane<true>(pw));
  } else {
    r = fmaf(pair_lane<false>(q.x), pair_lane<false>(p.x), r);
    r = fmaf(pair_lane<false>(q.y), pair_lane<false>(p.y), r);
    r = fmaf(pair_lane<false>(q.z), pair_lane<false>(p.z), r);
    r = fmaf(pair_lane<false>(q.w), pair_lane<false>(p.w), r);
    r = fmaf(pair_lane<true>(q.x), pair_lane<true>(p.x), r);
    r = fmaf(pair_lane<true>(q.y), pair_lane<true>(p.y), r);
    r = fmaf(pair_lane<true>(q.z), pair_lane<true>(p.z), r);
    r = fmaf(pair_lane<true>(q.w), pair_lane<true>(p.w), r);
  }
  return r;
}

// NP = pair steps (ceil(ceil(d/4) / 2)); rows and the staged query are zero padded to 16 floats, so a step may
// run past d: the extra terms are +0 and leave the (never negative-zero) running sum unchanged.
template <int NP>
__device__ __forceinline__ float mips_pair_ct(const float *__restrict__ prow, const float *qv, int d, int h) {
  const int tail_from = (d & ~7) >> 3;  // first fused step
  float4 buf[NP];
#pragma unroll
  for (int t = 0; t < NP; t++) buf[t] = *reinterpret_cast<const float4 *>(prow + 8 * t + 4 * h);
  float r = 0.f;
#pragma unroll
  for (int t = 0; t < NP; t++) {
    const float4 q = *reinterpret_cast<const float4 *>(qv + 8 * t + 4 * h);
    r = mips_step(r, buf[t], q, t >= tail_from);
  }
  return -r;
}

// two candidates per lane pair, every load of both rows in flight before the first use
template <int NP>
__device__ __forceinline__ void mips_pair2_ct(const float *__restrict__ prow0, const float *__restrict__ prow1,
                                              const float *qv, int d, int h, float &d0, float &d1) {
  const int tail_from = (d & ~7) >> 3;
  float4 b0[NP], b1[NP];
#pragma unroll
  for (int t = 0; t < NP; t++) b0[t] = *reinterpret_cast<const float4 *>(prow0 + 8 * t + 4 * h);
#pragma unroll
  for (int t = 0; t < NP; t++) b1[t] = *reinterpret_cast<const float4 *>(prow1 + 8 * t + 4 * h);
  float r0 = 0.f, r1 = 0.f;
#pragma unroll
  for (int t = 0; t < NP; t++) {
    const float4 q = *reinterpret_cast<const float4 *>(qv + 8 * t + 4 * h);
    const bool tail = t >= tail_from;
    r0 = mips_step(r0, b0[t], q, tail);
    r1 = mips_step(r1, b1[t], q, tail);
  }
  d0 = -r0;
  d1 = -r1;
}

// any dimension: blocks of 8 pair steps
__device__ __forceinline__ float mips_pair(const float *__restrict__ prow, const float *qv, int d, int h) {
  const int np = (((d + 3) >> 2) + 1) >> 1, tail_from = (d & ~7) >> 3;
  float r = 0.f;
  for (int t0 = 0; t0 < np; t0 += 8) {
    float4 buf[8];
#pragma unroll
    for (int j = 0; j < 8; j++)
      if (t0 + j < np) buf[j] = *reinterpret_cast<const float4 *>(prow + 8 * (t0 + j) + 4 * h);
#pragma unroll
    for (int j = 0; j < 8; j++)
      if (t0 + j < np) {
        const float4 q = *reinterpret_cast<const float4 *>(qv + 8 * (t0 + j) + 4 * h);
        r = mips_step(r, buf[j], q, t0 + j >= tail_from);
      }
  }
  return -r;
}

// --------------------------------------------------------------------------------------------
// uint8 / int8 point sets (WANN_DT = 1 / 2: one translation unit per element type, see wann_kernels_u8.hip):
// rows are BYTES (d elements, zero padded to a multiple of 64), `stride` still counts 32-bit words, distances
// are exact int32 sums cast to float like the reference's (euclidian_point.h:44-60, mips_point.h:44-58) for any
// dimension.  A lane pair owns a candidate; lane h takes the 16-byte chunks 2t + h; v_dot4 does four
// multiply-adds per instruction; squared L2 = sum a^2 + sum q^2 - 2 sum a q (each sum an exact integer).
// Both lanes of the pair return the distance.
// --------------------------------------------------------------------------------------------
#ifndef WANN_DT
#define WANN_DT 0  // 0 = float32 rows, 1 = uint8, 2 = int8
#endif
#if WANN_DT != 0
__device__ __forceinline__ int dot4_acc(uint32_t a, uint32_t b, int c) {
#if WANN_DT == 1
  return (int)__builtin_amdgcn_udot4(a, b, (uint32_t)c, false);
#else
  return __builtin_amdgcn_sdot4((int)a, (int)b, c, false);
#endif
}

template <int METRIC>
__device__ __forceinline__ float byte_pair(const float *__restrict__ prow, const float *qv, int stride_words, int h) {
  const int chunks = stride_words >> 3;  // 16-byte chunks per lane (the row has stride_words / 4 of them)
  int ab = 0, aa = 0, qq = 0;
  for (int t0 = 0; t0 < chunks; t0 += 8) {
    uint4 buf[8];
#pragma unroll
    for (int j = 0; j < 8; j++)
      if (t0 + j < chunks) buf[j] = *reinterpret_cast<const uint4 *>(prow + 4 * (2 * (t0 + j) + h));
#pragma unroll
    for (int j = 0; j < 8; j++)
      if (t0 + j < chunks) {
        const uint4 q = *reinterpret_cast<const uint4 *>(qv + 4 * (2 * (t0 + j) + h));
        const uint4 p = buf[j];
        ab = dot4_acc(p.x, q.x, ab);
        ab = dot4_acc(p.y, q.y, ab);
        ab = dot4_acc(p.z, q.z, ab);
        ab = dot4_acc(p.w, q.w, ab);
        if (METRIC == 0) {
          aa = dot4_acc(p.x, p.x, aa);
          aa = dot4_acc(p.y, p.y, aa);
          aa = dot4_acc(p.z, p.z, aa);
          aa = dot4_acc(p.w, p.w, aa);
          qq = dot4_acc(q.x, q.x, qq);
          qq = dot4_acc(q.y, q.y, qq);
          qq = dot4_acc(q.z, q.z, qq);
          qq = dot4_acc(q.w, q.w, qq);
        }
      }
  }
  const int mine = METRIC == 0 ? (aa + qq - 2 * ab) : ab;
  const int both = mine + __shfl_xor(mine, 1);
  return METRIC == 0 ? (float)both : -(float)both;
}

// the query's elements (integer-valued floats) packed four to a word, as the rows are
__device__ __forceinline__ float pack_query_word(const float *q, int64_t base, int i, int d) {
  uint32_t w = 0;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int e = 4 * i + j;
    if (e < d) w |= ((uint32_t)(int)q[base + e] & 0xffu) << (8 * j);
  }
  return __uint_as_float(w);
}
#endif

// staged query word i of `stride` (zero padded): a float for float32 rows, four packed elements for byte rows
__device__ __forceinline__ float stage_query_word(const float *queries, int64_t qrow, int i, int d) {
#if WANN_DT == 0
  return (i < d) ? queries[qrow * d + i] : 0.f;
#else
  return pack_query_word(queries, qrow * d, i, d);
#endif
}

// Pull the cache lines of rows ids_lds[first..first+count) towards the L2 without consuming them:
// one dword per 128-B line, result unused (the loads of the rows that follow then hit the cache
// instead of paying a second serial HBM round trip).
__device__ __forceinline__ void wave_touch_rows(const IndexView &ix, const int32_t *ids_lds, int first, int count,
                                                int64_t row_off) {
  const int lpr = (ix.stride * 4 + 127) >> 7;  // 128-B lines per row
  const int total = count * lpr;
  for (int t = lane_id(); t < total; t += 64) {
    const int c = first + t / lpr, line = t % lpr;
    const float *addr = ix.points + (row_off + ids_lds[c]) * (int64_t)ix.stride + line * 32;
    (void)*reinterpret_cast<const volatile int *>(addr);
  }
}

// Distances of `cnt` rows whose (sorted-order) row numbers sit in ids_lds[0..cnt): afterwards lane
// s < cnt holds the distance of row s.  scratch_lds: 64 floats.
// MIPS_TWO: the inner-product routine that keeps two rows per lane pair in flight (the exact scans: rows stream, registers
// are plentiful there); the search kernels use the one-row routine (see below).
template <int METRIC, bool TWO_ROWS = true, bool MIPS_TWO = false>
__device__ __forceinline__ float wave_distances(const IndexView &ix, const int32_t *ids_lds,
                                                float *scratch_lds, const float *qv, int cnt,
                                                int64_t row_off) {
  const int lane = lane_id();
#if WANN_DT != 0
  {
    const int h = lane & 1;
    for (int base = 0; base < cnt; base += 32) {
      const int s = base + (lane >> 1);
      const bool act = s < cnt;
      const int id = act ? ids_lds[s] : 0;
      const float dd = byte_pair<METRIC>(ix.points + (row_off + id) * (int64_t)ix.stride, qv, ix.stride, h);
      if (act && h) scratch_lds[s] = dd;
    }
    WAVE_SYNC();
    const float r = (lane < cnt) ? scratch_lds[lane] : 0.f;
    WAVE_SYNC();
    return r;
  }
#endif
  if (METRIC == 1) {
    // One row per lane pair and pass (up to two passes; the second pass's cache lines are requested before the first is
    // scored).  Keeping two rows per pair in flight -- one round trip for up to 64 candidates -- costs 52 more registers, and
    // the four-wave inner-product kernel runs three waves per SIMD (168 registers): more searches in flight beat the shorter
    // first hops of each.
    const int h = lane & 1;
    const int np = (((ix.d + 3) >> 2) + 1) >> 1;  // wave-uniform
    if (MIPS_TWO && cnt > 32 && (np == 12 || np == 13)) {  // one round trip for up to 64 rows
      const int s0 = lane >> 1, s1 = 32 + (lane >> 1);
      const bool act1 = s1 < cnt;
      const int id0 = ids_lds[s0], id1 = act1 ? ids_lds[s1] : 0;
      const float *p0 = ix.points + (row_off + id0) * (int64_t)ix.stride;
      const float *p1 = ix.points + (row_off + id1) * (int64_t)ix.stride;
      float d0, d1;
      if (np == 12) mips_pair2_ct<12>(p0, p1, qv, ix.d, h, d0, d1);
      else mips_pair2_ct<13>(p0, p1, qv, ix.d, h, d0, d1);
      if (h) scratch_lds[s0] = d0;
      if (act1 && h) scratch_lds[s1] = d1;
      WAVE_SYNC();
      float r = (lane < cnt) ? scratch_lds[lane] : 0.f;
      WAVE_SYNC();
      return r;
    }
    if (cnt > 32) wave_touch_rows(ix, ids_lds, 32, cnt - 32, row_off);
    for (int base = 0; base < cnt; base += 32) {
      const int s = base + (lane >> 1);
      const bool act = s < cnt;
      const int id = act ? ids_lds[s] : 0;  // idle pairs score node 0: no branches
      const float *prow = ix.points + (row_off + id) * (int64_t)ix.stride;
      float dd;
      switch (np) {
        case 12: dd = mips_pair_ct<12>(prow, qv, ix.d, h); break;  // d = 96
        case 13: dd = mips_pair_ct<13>(prow, qv, ix.d, h); break;  // d = 100
        default: dd = mips_pair(prow, qv, ix.d, h); break;
      }
      if (act && h) scratch_lds[s] = dd;
    }
    WAVE_SYNC();
    float r = (lane < cnt) ? scratch_lds[lane] : 0.f;
    WAVE_SYNC();
    return r;
  } else {
    const int D8 = (ix.d + 7) >> 3;
    const int h = lane & 1;
    if (TWO_ROWS && (D8 == 16 || D8 == 12)) {  // d = 128 / 96: one round trip for up to 64 candidates
      const int s0 = lane >> 1, s1 = 32 + (lane >> 1);
      const bool act0 = s0 < cnt, act1 = s1 < cnt;
      const int id0 = act0 ? ids_lds[s0] : 0, id1 = act1 ? ids_lds[s1] : 0;
      const float *p0 = ix.points + (row_off + id0) * (int64_t)ix.stride;
      const float *p1 = ix.points + (row_off + id1) * (int64_t)ix.stride;
      float d0, d1;
      if (cnt > 32) {
        if (D8 == 16) l2_pair2_ct<16>(p0, p1, qv, h, d0, d1);
        else l2_pair2_ct<12>(p0, p1, qv, h, d0, d1);
        if (act1 && h) scratch_lds[s1] = d1;
      } else {
        if (D8 == 16) d0 = l2_pair_ct<16>(p0, qv, h, act0);
        else d0 = l2_pair_ct<12>(p0, qv, h, act0);
      }
      if (act0 && h) scratch_lds[s0] = d0;
    } else {
      if (cnt > 32) wave_touch_rows(ix, ids_lds, 32, cnt - 32, row_off);  // second pass: start its misses now
      for (int base = 0; base < cnt; base += 32) {
        int s = base + (lane >> 1);
        bool act = s < cnt;
        int id = act ? ids_lds[s] : 0;
        const float *prow = ix.points + (row_off + id) * (int64_t)ix.stride;
        float dist;
        switch (D8) {  // wave-uniform
          case 16: dist = l2_pair_ct<16>(prow, qv, h, act); break;  // d = 128 (TWO_ROWS = false)
          case 12: dist = l2_pair_ct<12>(prow, qv, h, act); break;  // d = 96
          case 13: dist = l2_pair_ct<13>(prow, qv, h, act); break;  // d = 100
          case 8: dist = l2_pair_ct<8>(prow, qv, h, act); break;    // d = 64
          default: dist = l2_pair<16>(prow, qv, D8, h, act); break;
        }
        if (act && h) scratch_lds[s] = dist;
      }
    }
    WAVE_SYNC();
    float r = (lane < cnt) ? scratch_lds[lane] : 0.f;
    WAVE_SYNC();
    return r;
  }
}

// --------------------------------------------------------------------------------------------
// sorted-list merge: insert the candidates flagged `pass` (key = fkey(dist)<<32 | id<<1) into the
// sorted list beam[0..m) of capacity B, dropping candidates already present (same id and dist),
// exactly like std::set_union + truncate (beamSearch.h:148-157).  Bit 0 of an entry is its
// "visited" flag and is ignored by comparisons.  Returns the new size; *first_pos receives the
// position of the first inserted element (or the old size when nothing was inserted).
// cand_key: 64 u64 of per-wave LDS scratch.
// --------------------------------------------------------------------------------------------
// PRESORTED: the candidates sit in lanes 0..c-1 in ascending key order already (no ranking pass).
// prev_key / prev_mask (rows of more than 64 neighbours are worked in two halves, wave_beam_search): the passing candidates of the
// row's FIRST half, in their lanes -- a copy of one of them among this call's candidates counts as a further copy of the same
// std::set_union operand (the reference unions the whole row's candidates at once).
// COLLAPSE (wave_beam_search_mid: the candidates of SEVERAL hops, none of which holds a key twice, united in one go): hop after hop
// the reference's union would keep max(copies in the beam, copies in the hop) of a key -- so a key the beam holds is dropped and
// equal candidates (the same node scored in two hops) count once.
template <typename BeamPtr, bool DEDUP = true, bool PRESORTED = false, bool COLLAPSE = false>
__device__ __forceinline__ int wave_merge(BeamPtr beam, int m, int B, bool pass, u64 key,
                                          u64 *cand_key, int *first_pos, u64 prev_key = 0ull, u64 prev_mask = 0ull) {
  const int lane = lane_id();
  *first_pos = m;
  u64 smask = ballot64(pass);
  if (smask == 0) return m;
  int rank = PRESORTED ? lane : 0;
  if (!PRESORTED)
    for (u64 mm = smask; mm; mm &= mm - 1) {
      int l = ctz64(mm);
      u64 kl = rdlane64(key, l);
      rank += (kl < key || (kl == key && l < lane)) ? 1 : 0;
    }
  if (pass) cand_key[rank] = key;
  WAVE_SYNC();
  const int c = popc64(smask);
  const bool mine = lane < c;
  u64 ck = mine ? cand_key[lane] : ~0ull;
  // lower bound of ck in beam[0..m)
  int pos;
  if (c <= 16 && m > 64) {
    // K-ary search: with few candidates (the usual case once the beam is full) K = 64 / c' lanes serve one
    // candidate and probe K-1 pivots of its interval per step -- about log_K(m) dependent beam reads instead
    // of log_2(m).  Invariant per group: entries below lo are < key, entries from hi on are >= key.
    const int logK = (c <= 1) ? 6 : (c <= 2) ? 5 : (c <= 4) ? 4 : (c <= 8) ? 3 : 2;
    const int K = 1 << logK;
    const int g = lane >> logK, sub = lane & (K - 1);
    const u64 gk = (g < c ? cand_key[g] : ~0ull) | 1ull;
    const u64 gmask = (K == 64) ? ~0ull : (((u64)1 << K) - 1);
    int lo = 0, hi = m;
    for (;;) {
      const bool open = lo < hi;
      if (ballot64(open) == 0) break;
      const int width = hi - lo;
      bool less = false;
      if (open && sub > 0) less = (beam[lo + ((sub * width) >> logK)] | 1ull) < gk;
      const int t = popc64((ballot64(less) >> (g << logK)) & gmask);  // pivots 1..t are < key
      if (open) {
        const int qt = lo + ((t * width) >> logK), qn = lo + (((t + 1) * width) >> logK);
        if (t < K - 1) hi = qn;
        if (t >= 1) lo = qt + 1;
      }
    }
    pos = __shfl(lo, (lane << logK) & 63);  // lane i < c: the result of group i
  } else {
    int lo = 0, hi = m;
    const int iters = 32 - __builtin_clz(m | 1) + 1;
    for (int it = 0; it < iters; it++) {
      if (lo < hi) {
        int mid = (lo + hi) >> 1;
        u64 bv = beam[mid] | 1ull;
        if (bv < (ck | 1ull)) lo = mid + 1;
        else hi = mid;
      }
    }
    pos = lo;
  }
  // std::set_union keeps max(copies in beam, copies among candidates) of equal elements: the j-th
  // copy of a candidate key is dropped iff the beam already holds more than j copies.  (Copies
  // arise when a row lists a node twice -- the reference's builder can append the start point
  // twice -- and the lossy filter lets both through.)
  bool dup = false;
  int prior = 0;  // copies of this key among the first half's candidates (wave-uniform loop; nearly always empty)
  if (DEDUP)
    for (u64 mm = prev_mask; mm; mm &= mm - 1) prior += (mine && (rdlane64(prev_key, ctz64(mm)) | 1ull) == (ck | 1ull)) ? 1 : 0;
  if (DEDUP && mine) {
    int j = prior, bx = 0;
    for (int l = lane - 1; l >= 0 && cand_key[l] == ck; l--) j++;
    while (pos + bx < m && ((beam[pos + bx] | 1ull) == (ck | 1ull))) bx++;
    dup = COLLAPSE ? (j > 0 || bx > 0) : (j < bx);
  }
  WAVE_SYNC();
  const u64 nd = ballot64(mine && !dup);
  const int cp = popc64(nd);
  if (cp == 0) return m;
  const int pre = popc64(nd & lanemask_lt());
  const int p0 = rdlane(pos, ctz64(nd));
  const int span = m - p0;
  if (span > 0) {
    // Shift beam[p0..m) right, highest chunk first, four 64-entry chunks per step (all reads of a step are
    // issued before its first write: entries only move right, by at most cp <= 64).  An entry moves by the
    // number of inserted candidates whose position is <= its index: the candidates at or before the
    // chunk's first index shift the whole chunk (one ballot), the few landing inside it a suffix.
    const bool ndl = mine && !dup;
    // (a presorted list is the one-wave kernel's delta list going into a beam of thousands: eight chunks per step -- the two
    // LDS round trips of a step are what the shift waits for, and that kernel has the registers)
    constexpr int NCH = PRESORTED ? 8 : 4;
    for (int top = p0 + ((span - 1) & ~63); top >= p0; top -= 64 * NCH) {
      u64 ev[NCH];
      int nx[NCH];
#pragma unroll
      for (int j = 0; j < NCH; j++) {
        const int base = top - 64 * j;
        nx[j] = B;  // "do not write"
        ev[j] = 0ull;
        if (base >= p0) {  // wave-uniform
          const int x = base + lane;
          const bool act = x < m;
          if (act) ev[j] = beam[x];
          int sx = popc64(ballot64(ndl && pos <= base));
          for (u64 mm = ballot64(ndl && pos > base && pos <= base + 63); mm; mm &= mm - 1) {
            const int pl = rdlane(pos, ctz64(mm));
            sx += (pl <= x) ? 1 : 0;
          }
          if (act) nx[j] = x + sx;
        }
      }
      WAVE_SYNC();
#pragma unroll
      for (int j = 0; j < NCH; j++)
        if (nx[j] < B) beam[nx[j]] = ev[j];
      WAVE_SYNC();
    }
  }
  if (mine && !dup) {
    int np = pos + pre;
    if (np < B) beam[np] = ck;
  }
  WAVE_SYNC();
  *first_pos = p0;
  int nm = m + cp;
  return nm < B ? nm : B;
}


// --------------------------------------------------------------------------------------------
// A sorted run of D <= 64 NEW keys (lane j holds the j-th smallest; none is in the beam yet) into the sorted LDS beam
// mb[0..M): the one-wave kernel's delta list going into a beam of thousands, every fourteen hops of a long search.  Worked
// in DESTINATION space, where the run's slots are distinct positions (slot of key j = its lower bound + j): a bit per
// position in `flagw` (LDS words, one ds_or for the whole run), and for a chunk of 64 destinations the number of run slots
// below destination y is a running count plus a v_mbcnt of the chunk's 64 flag bits -- no loop over the keys that land in
// a chunk, no rank arithmetic: twenty instructions per chunk where the general union (wave_merge) spends forty-five.
// Chunks go top down; a chunk's sources lie at or below its destinations and above nothing that has been written yet, and
// LDS operations of a wave execute in order, so reads and writes of neighbouring chunks need no waits between them.
// Returns M + D; *first_pos = position of the first key of the run.  flagw must hold (M + D + 127) / 32 words.
// --------------------------------------------------------------------------------------------
__device__ __forceinline__ int wave_merge_run(u64 *mb, int M, u64 dk, int D, int32_t *flagw, int *first_pos) {
  const int lane = lane_id();
  const u64 k1 = dk | 1ull;  // (lanes >= D hold ~0: their lower bound is M)
  int lo = 0, hi = M;
  const int iters = 32 - __builtin_clz(M | 1) + 1;
  for (int it = 0; it < iters; it++) {
    const int mid = (lo + hi) >> 1;
    const u64 bv = mb[mid < M ? mid : (M > 0 ? M - 1 : 0)] | 1ull;
    if (lo < hi) {
      if (bv < k1) lo = mid + 1;
      else hi = mid;
    }
  }
  const int posd = lo + lane;  // destination of this lane's key
  const int newM = M + D;
  const int p0 = rdlane(lo, 0);
  const int base_lo = p0 & ~63;
  for (int w = (base_lo >> 5) + lane; w <= ((newM - 1) >> 5) + 1; w += 64) flagw[w] = 0;
  WAVE_SYNC();
  if (lane < D) __hip_atomic_fetch_or(&flagw[posd >> 5], (int32_t)(1u << (posd & 31)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  WAVE_SYNC();
  int H = D;  // run slots below the end of the chunk being worked
  constexpr int NCH = 4;
  for (int top = (newM - 1) & ~63; top >= base_lo; top -= 64 * NCH) {
    u64 cm[NCH];
#pragma unroll
    for (int j = 0; j < NCH; j++) {  // (no branches around the reads: a chunk below the first moved one reads that one's flags and drops them)
      const int base = top - 64 * j;
      cm[j] = *reinterpret_cast<const u64 *>(flagw + ((base >= base_lo ? base : base_lo) >> 5));  // (the same word pair in every lane)
    }
    u64 ev[NCH];
    int dst[NCH];
#pragma unroll
    for (int j = 0; j < NCH; j++) {
      const int base = top - 64 * j;
      u64 c = (u64)uni64((long long)cm[j]);
      if (base < base_lo) c = 0ull;
      const int H0 = H - popc64(c);
      H = H0;
      const int y = base + lane;
      const int h = H0 + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(c >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)c, 0u));
      const bool mv = y < newM && y >= p0 && !((c >> lane) & 1ull);  // (y >= p0 >= base_lo: never in a chunk below the first)
      dst[j] = mv ? y : -1;
      ev[j] = mb[mv ? y - h : 0];
    }
#pragma unroll
    for (int j = 0; j < NCH; j++)
      if (dst[j] >= 0) mb[dst[j]] = ev[j];
  }
  if (lane < D) mb[posd] = dk;
  WAVE_SYNC();
  *first_pos = p0;
  return newM;
}

// --------------------------------------------------------------------------------------------
// beam-search core: one wavefront runs one search (beamSearch.h:51-184) over partition `part`
// for the query staged (zero padded) in L.qv.  The beam lives in LDS (BEAM_LDS) or in gbeam, the
// seen-filter in LDS (TABLE_LDS) or in gtable.  COLLECT appends every visited (dist,id) key to
// vis[] in visit order (the build needs the visited list, vamana/index.h:270-272).
// --------------------------------------------------------------------------------------------
struct WaveLds {
  float *qv;
  u64 *cand_key;
  int32_t *cand_id;
  float *cand_dist;
  u64 *lbeam;
  int32_t *ltable;
};

__device__ __forceinline__ int wave_lds_common_bytes(int stride) {
  return ((stride * 4 + 15) & ~15) + 64 * 8 + 64 * 4 + 64 * 4;
}

__device__ __forceinline__ WaveLds carve_wave_lds(unsigned char *base, int stride, int B, bool beam_lds) {
  WaveLds L;
  L.qv = reinterpret_cast<float *>(base);
  int off = (stride * 4 + 15) & ~15;
  L.cand_key = reinterpret_cast<u64 *>(base + off);
  off += 64 * 8;
  L.cand_id = reinterpret_cast<int32_t *>(base + off);
  off += 64 * 4;
  L.cand_dist = reinterpret_cast<float *>(base + off);
  off += 64 * 4;
  L.lbeam = reinterpret_cast<u64 *>(base + off);
  if (beam_lds) off += ((B + 1) & ~1) * 8;
  L.ltable = reinterpret_cast<int32_t *>(base + off);
  return L;
}

template <int METRIC, bool TABLE_LDS, bool BEAM_LDS, bool COLLECT, bool CUT = false>
__device__ __forceinline__ void wave_beam_search(const IndexView &ix, const PartDesc &part, const WaveLds &L,
                                                 u64 *gbeam, int32_t *gtable, int B, int bits, int64_t qid,
                                                 int64_t limit, int degree_limit, u64 *vis, int vis_cap,
                                                 int &m_out, long long &nvis_out, long long &ncmp_out,
                                                 unsigned long long *prof = nullptr, int32_t *mini = nullptr,
                                                 uint32_t mini_mask = 0, int cut_k = 0, double cut = 0.0,
                                                 uint32_t *vset = nullptr) {
  prof = WANN_PROF_PTR(prof);
  if (!CUT) {  // (compile-time: the post-filter path never takes the cut step and keeps no scalar state for it)
    cut_k = 0;
    vset = nullptr;
  }
  const int lane = lane_id();
  const uint32_t tmask = (1u << bits) - 1u;
  const int64_t row_off = part.start;
  if (TABLE_LDS) {  // (16-byte aligned: carve_wave_lds)
    int4 *lt = reinterpret_cast<int4 *>(L.ltable);
    for (int i = lane; i < (1 << (bits - 2)); i += 64) lt[i] = make_int4(-1, -1, -1, -1);
  } else {
    int4 *gt = reinterpret_cast<int4 *>(gtable);
    for (int i = lane; i < (1 << (bits - 2)); i += 64) gt[i] = make_int4(-1, -1, -1, -1);
  }
  // With the cut step a VISITED entry can leave the beam and be admitted again later (the beam is no longer always full,
  // so the cutoff is not monotone): the reference keeps its visited list (beamSearch.h:114-116,175-178), here the exact
  // set of visited nodes is a bitmap and a re-admitted node gets its visited bit back.  (Without the cut an evicted node
  // can never return and the bit in the beam entry is all that is needed.)
  if (vset) {
    int4 *sv = reinterpret_cast<int4 *>(vset);
    const int n16 = (part.n + 127) >> 7;
    for (int i = lane; i < n16; i += 64) sv[i] = make_int4(0, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  // frontier = {start node 0} (beamSearch.h:80-82)
  if (lane == 0) L.cand_id[0] = 0;
  WAVE_SYNC();
  float d0 = wave_distances<METRIC>(ix, L.cand_id, L.cand_dist, L.qv, 1, row_off);
  d0 = __shfl(d0, 0);
  int m = 1, p = 0;
  long long nvis = 0, ncmp = 1;
  auto beam_ld = [&](int i) -> u64 { return BEAM_LDS ? L.lbeam[i] : gbeam[i]; };
  auto beam_st = [&](int i, u64 v) {
    if (BEAM_LDS) L.lbeam[i] = v;
    else gbeam[i] = v;
  };
  if (lane == 0) beam_st(0, ((u64)fkey(d0) << 32));
  WAVE_SYNC();

  // optional per-phase cycle accounting (dev tool: wann_raw_beam_search with a profile buffer)
  unsigned long long tp = 0, acc[6] = {0, 0, 0, 0, 0, 0};
#define WANN_PHASE(i)                                    \
  do {                                                   \
    if (prof) {                                          \
      unsigned long long tn = __builtin_readcyclecounter(); \
      acc[i] += tn - tp;                                 \
      tp = tn;                                           \
    }                                                    \
  } while (0)
  if (prof) tp = __builtin_readcyclecounter();
  while (p < m && nvis < limit) {
    // ---- visit the closest unvisited beam entry (beamSearch.h:111-117)
    const u64 curkey = beam_ld(p);
    const int cur = (int)((uint32_t)curkey >> 1);
    if (lane == 0) {
      beam_st(p, curkey | 1ull);
      if (COLLECT && nvis < vis_cap) vis[nvis] = curkey & ~1ull;
      if (vset) __hip_atomic_fetch_or(vset + (cur >> 5), 1u << (cur & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    nvis++;

    // ---- the cutoff of this hop (beamSearch.h:135-137): fixed BEFORE the row's candidates enter the beam
    float cutoff = 2147483648.0f;  // (float)INT_MAX
    if (m >= B) cutoff = funkey((uint32_t)(beam_ld(m - 1) >> 32));
    // A row of more than 64 neighbours (64 < R <= 128, graph.h:115-124 takes any R) is worked in two halves of 64 lanes: the
    // lossy filter is sequential over the row's elements anyway (the second half sees the table the first left), both halves
    // are scored against the hop's cutoff, and the union of (beam + first half, truncated) with the second half keeps the
    // same B smallest entries as one union of everything; std::set_union's multiset rule needs the first half's candidates
    // when the second half's are merged (wave_merge: prev_key / prev_mask).
    const int nhalf = (ix.rs + 63) >> 6;
    int p0 = m;
    u64 key_h0 = 0ull, mask_h0 = 0ull;
    for (int half = 0; half < nhalf; half++) {
    // ---- adjacency row, coalesced (graph.h:198); -1 = unused slot
    const int slot = 64 * half + lane;
    int a = -1;
    if (slot < ix.rs) a = ix.graph[(part.row_base + cur) * (int64_t)ix.rs + slot];
    bool valid = (a >= 0) && (slot < degree_limit) && ((int64_t)a != qid);
    if (prof) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    WANN_PHASE(0);  // row fetch

    // ---- lossy direct-mapped "seen" filter, sequential semantics emulated exactly
    //      (beamSearch.h:68-73,126-131): lane i sees the id left in its slot by the nearest
    //      preceding lane of the row that hashed to the same slot, else the table's old value;
    //      the last lane of each slot class leaves its id in the table.
    const uint32_t loc = (uint32_t)hash64_2((u64)(uint32_t)a) & tmask;
    int old = -1;
    if (valid) old = TABLE_LDS ? L.ltable[loc] : gtable[loc];
    // cheap necessary condition for "two lanes of the row share a filter slot": they would also share a
    // slot of a small LDS scratch hash.  No clash there => the sequential rule is just "old == id".
    bool may_clash = true;
    if (mini) {
      const uint32_t h = loc & mini_mask;
      if (valid) mini[h] = lane;
      WAVE_SYNC();
      const bool c = valid && (mini[h] != lane);
      may_clash = ballot64(c) != 0;
      WAVE_SYNC();
    }
    bool seen;
    if (!may_clash) {
      seen = valid && (old == a);
      if (valid) {
        if (TABLE_LDS) L.ltable[loc] = a;
        else gtable[loc] = a;
      }
    } else {
      u64 eq = ballot64(valid);
      for (int b = 0; b < bits; b++) {
        bool bit = (loc >> b) & 1u;
        u64 bm = ballot64(valid && bit);
        eq &= bit ? bm : ~bm;
      }
      const u64 lower = eq & lanemask_lt();
      const u64 higher = (lane == 63) ? 0ull : (eq >> (lane + 1));
      int prev_lane = lower ? (63 - __builtin_clzll(lower)) : lane;
      int prev_val = __shfl(a, prev_lane);
      if (!lower) prev_val = old;
      seen = valid && (prev_val == a);
      WAVE_SYNC();
      if (valid && higher == 0) {
        if (TABLE_LDS) L.ltable[loc] = a;
        else gtable[loc] = a;
      }
    }
    const bool keep = valid && !seen;
    const u64 kmask = ballot64(keep);
    const int nk = popc64(kmask);
    if (keep) L.cand_id[popc64(kmask & lanemask_lt())] = a;
    WAVE_SYNC();
    ncmp += nk;
    WANN_PHASE(1);  // seen-filter

    // ---- score the kept neighbours (beamSearch.h:135-145)
    float dist = wave_distances<METRIC>(ix, L.cand_id, L.cand_dist, L.qv, nk, row_off);
    int cid = (lane < nk) ? L.cand_id[lane] : 0;
    WAVE_SYNC();
    const bool pass = (lane < nk) && (dist < cutoff);
    u64 key = ((u64)fkey(dist) << 32) | ((u64)(uint32_t)cid << 1);
    if (vset && pass && ((__hip_atomic_load(vset + (cid >> 5), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> (cid & 31)) & 1u)) key |= 1ull;
    WANN_PHASE(2);  // vector fetch + distances

    // ---- sort + set_union + truncate (beamSearch.h:148-157)
    int p0h;
    if (BEAM_LDS) m = wave_merge(L.lbeam, m, B, pass, key, L.cand_key, &p0h, key_h0, mask_h0);
    else m = wave_merge(gbeam, m, B, pass, key, L.cand_key, &p0h, key_h0, mask_h0);
    p0 = p0h < p0 ? p0h : p0;
    key_h0 = key;
    mask_h0 = ballot64(pass);
    }  // (the row's halves)
    // ---- beamSearch.h:159-167: with a k (unfiltered VamanaIndex queries) and a metric distance, entries beyond
    //      cut * (distance of entry k) leave: upper_bound of (id 0, that distance) under (dist, id) order
    if (METRIC == 0 && cut_k > 0 && m > cut_k) {
      const float thr = (float)(cut * (double)funkey((uint32_t)(beam_ld(cut_k) >> 32)));
      int keep = 0;
      for (int s0 = 0; s0 < m; s0 += 64) {
        const int x = s0 + lane;
        const u64 ev = x < m ? beam_ld(x) : ~0ull;
        const float dv = funkey((uint32_t)(ev >> 32));
        const bool stay = x < m && (dv < thr || (dv == thr && ((uint32_t)ev >> 1) == 0u));
        const u64 sm = ballot64(stay);
        keep += popc64(sm);
        if (sm != ~0ull) break;  // (sorted: the entries that stay are a prefix)
      }
      m = keep;
      if (p0 > m) p0 = m;
    }
    WANN_PHASE(3);  // sort + merge

    // ---- next = first beam entry not yet visited (beamSearch.h:175-178)
    int sp = p < p0 ? p : p0;
    p = m;
    while (sp < m) {
      int x = sp + lane;
      bool un = (x < m) && !(beam_ld(x) & 1ull);
      u64 bm = ballot64(un);
      if (bm) {
        p = sp + ctz64(bm);
        break;
      }
      sp += 64;
    }
    WANN_PHASE(4);  // next-node scan
  }
#undef WANN_PHASE
  if (prof && lane == 0)
    for (int i = 0; i < 5; i++) atomicAdd(&prof[i], acc[i]);
  m_out = m;
  nvis_out = nvis;
  ncmp_out = ncmp;
}


// --------------------------------------------------------------------------------------------
// General beam-search core for beams that do not fit the register-resident variant (B > 128): the same
// search as wave_beam_search<.., false, true, false> (beam in the LDS, lossy seen-filter in global memory)
// with the three per-hop costs of that routine removed from the chain of dependent hops:
//
//  * EXACT SEEN SET.  Besides the reference's lossy filter (beamSearch.h:66-73, kept bit for bit: it decides
//    which neighbours the reference scores, i.e. the dist_cmps counter) the search keeps an exact bitmap of
//    the nodes it has scored.  Scoring a node a second time can never change the beam: its distance is the
//    same; either it is still in the beam (std::set_union drops the copy, :151-154) or it was cut off / never
//    admitted at a cutoff that has only decreased since (:135-145), so `d < cutoff` fails again.  Re-scores
//    are therefore COUNTED (dist_cmps) but not computed, and every candidate that is computed is new: the
//    union needs no duplicate test.  (Rows that list a node twice are the one exception -- the reference's
//    multiset union can then keep two copies -- and take the old exact path.)
//  * DELTA LIST.  New entries go into a sorted 64-entry list in REGISTERS (one key per lane; a DPP lane
//    shift per insertion); the LDS beam is merged with it only when the list is full.  The beam is the
//    union of the two; it is kept truncated to B eagerly (the larger of the two last entries leaves), so
//    the cutoff is the larger of two scalar values.  No O(B) shift per hop.
//  * TAGGED FILTER ENTRIES.  A filter entry is (epoch << 24 | id): a search invalidates its predecessor's
//    entries by taking the next epoch instead of clearing up to 32 MiB.
//
// Results, hops and dist_cmps are those of wave_beam_search (the parity tests run both against the oracle).
// On return the whole beam is in L.lbeam[0..m).
// --------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 wave_shr1(u64 v) {  // lane i receives lane i-1's value (DPP wave_shr:1); lane 0 keeps its own
  const int lo = __builtin_amdgcn_update_dpp((int)(uint32_t)v, (int)(uint32_t)v, 0x138, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp((int)(uint32_t)(v >> 32), (int)(uint32_t)(v >> 32), 0x138, 0xf, 0xf, false);
  return ((u64)(uint32_t)hi << 32) | (uint32_t)lo;
}

// --------------------------------------------------------------------------------------------
// Scoring helper waves (one-wave-per-search kernel only).  A long search is one chain of dependent hops, each of
// which is three dependent memory round trips (adjacency row -> filter / seen-set probes -> vectors) plus the
// distance arithmetic.  Which node a hop visits is almost always known several hops ahead (the next unvisited
// entries of the beam), and row, filter slots and distances of a node are PURE functions of (query, node).  So the
// search wave REQUESTS the nodes it expects to visit and three helper waves of the same workgroup prepare a PACKET
// per node in the LDS: the node's adjacency row, its neighbours' filter slots, whether two of them share a slot, and
// the distances of the neighbours that are not in the exact seen set yet (they also touch the filter slots, so the
// search wave's own probes hit the L2).  The search wave takes all that from the packet; everything that has
// sequential semantics -- the lossy filter, the seen set, cutoff, union -- stays in the search wave, in program
// order.  A packet that is missing, late or lacks a distance costs the search wave the work it would have done
// anyway; a helper can not change a result.
// Request i (a node) is served by helper i % kHelpers into packet slot i % kPkSlots; the search wave keeps the
// slot -> node map in a register, so finding a packet costs no memory access.
// --------------------------------------------------------------------------------------------
// Scope of the exact seen set's probes and updates.  Only the waves of ONE workgroup (the search wave and its helper waves: one
// CU, one L1, one L2) ever touch a search's bitmap, so workgroup scope is enough -- and it matters: a device-scope atomic on a
// chip with one L2 per XCD is performed beyond the L2 (fabric / memory side, > 1 us), and since a wave's vector-memory counter
// retires in order every later load of the hop waited behind it.
constexpr int kSeenScope = __HIP_MEMORY_SCOPE_WORKGROUP;
// Packets are requested for the next WANN_BIG_LOOKAHEAD unvisited entries of the window (< kPkSlots - 2: request r + kPkSlots
// reuses the slot of request r).  Measured on one box, SIFT-1M-like 2^-9 batch / lone beam-5120 search: 4 -> 11.82 / 8.33 ms,
// 8 (with the filter store held back, see the fast path) -> 11.20 / 8.08 ms, 10 -> the same as 8.
#ifndef WANN_BIG_LOOKAHEAD
#define WANN_BIG_LOOKAHEAD 8
#endif
constexpr int kPkSlots = 12;
constexpr int kReqRing = 16;
static_assert(kPkSlots % kHelpers == 0 && kReqRing >= kPkSlots, "request -> helper / slot mapping");
struct ScoreBox {     // LDS mailbox; head written by the search wave, packet slot s by helper s % kHelpers
  int32_t gen;        // < 0: the kernel is ending; 0: no search is running; > 0: generation of the running search
  int32_t req_head;   // requests the running search has issued
  int32_t part;       // its partition
  int32_t bits;       // its filter size (log2)
  int32_t qid_lo, qid_hi;  // the query's own id (never scored, beamSearch.h:128)
  int32_t last_gen;   // the last generation used (generations never repeat within a launch)
  int32_t pad0;
  int32_t req[kReqRing];                // request i: the node, at i % kReqRing
  unsigned long long tag[kPkSlots];     // (generation << 32 | node) of a complete packet; 0 while a helper rewrites the slot
  unsigned long long mask[kPkSlots];    // lanes of the row whose distance the packet holds
  int32_t flags[kPkSlots];              // bit 0: two valid lanes of the row share a filter slot
  int32_t row[kPkSlots][64];            // adjacency row (-1: unused slot)
  uint32_t loc[kPkSlots][64];           // filter slot of each neighbour
  float dist[kPkSlots][64];
};
static_assert(sizeof(ScoreBox) <= kScoreBoxBytes, "ScoreBox outgrew its LDS reservation");
// The mailbox is polled, so its accesses are volatile -- through a pointer that carries the LDS address space: hipcc's
// address-space inference leaves volatile accesses through a generic pointer alone, and they become FLAT loads / stores
// (sc0 sc1) that wait for every outstanding global load of the wave.
typedef __attribute__((address_space(3))) volatile ScoreBox LdsBox;
__device__ __forceinline__ LdsBox *lds_box(ScoreBox *b) { return (LdsBox *)b; }

template <int METRIC, bool LEAN = false>
__device__ __forceinline__ float wave_distances_own(const IndexView &ix, int a, bool take, const float *qv, int64_t row_off);

template <int METRIC>
__device__ __forceinline__ void score_helper(const IndexView &ix, int32_t *gtable, const uint32_t *gseen, int degree_limit,
                                             ScoreBox *box, const float *qv, int hidx) {
  const int lane = lane_id();
  LdsBox *vb = lds_box(box);
  int my_gen = 0, next = hidx, bits = 10;
  int64_t row_base = 0, row_off = 0, qid = -1;
  uint32_t tmask = 0;
  for (;;) {
    const int gen = vb->gen;
    if (gen < 0) return;
    if (gen == 0) {
      my_gen = 0;
      __builtin_amdgcn_s_sleep(8);
      continue;
    }
    if (gen != my_gen) {  // a new search (its query is staged and its head fields are written before the generation)
      my_gen = gen;
      next = hidx;
      const PartDesc pd = ix.parts[vb->part];
      row_base = pd.row_base;
      row_off = pd.start;
      bits = vb->bits;
      tmask = (1u << bits) - 1u;
      qid = ((int64_t)vb->qid_hi << 32) | (uint32_t)vb->qid_lo;
    }
    const int head = vb->req_head;
    while (head - next > kReqRing - 4) next += kHelpers;  // far behind: those ring entries are gone, nobody waits for them
    if (next >= head) {
      __builtin_amdgcn_s_sleep(1);
      continue;
    }
    const int pnode = vb->req[next & (kReqRing - 1)];
    // (the search may have ended and the next one published requests between the reads above: `pnode` would then be a node
    // of ANOTHER partition, and the addresses formed below from this search's row_base / row_off could lie outside the
    // graph / point allocations -- the packet would be ignored (old generation), the loads would not be harmless)
    if (vb->gen != my_gen) continue;
    const int sl = next % kPkSlots;
    next += kHelpers;
    if (lane == 0) vb->tag[sl] = 0ull;  // readers of the old packet notice (this wave's LDS operations execute in order)
    int a = -1;
    if (lane < ix.rs) a = ix.graph[(row_base + pnode) * (int64_t)ix.rs + lane];
    const bool valid = (a >= 0) && (lane < degree_limit) && ((int64_t)a != qid);
    const uint32_t loc = (uint32_t)hash64_2((u64)(uint32_t)a) & tmask;
    uint32_t w = ~0u;
    int touch = 0;
    if (valid) {
      touch = gtable[loc];  // the search wave's filter probe will hit the L2
      w = __hip_atomic_load(gseen + (a >> 5), __ATOMIC_RELAXED, kSeenScope);
    }
    // do two valid lanes of the row share a filter slot?  (exact, by `bits` ballots: this wave is not on the critical path)
    u64 eq = ballot64(valid);
    for (int b = 0; b < bits; b++) {
      const bool bit = (loc >> b) & 1u;
      const u64 bm = ballot64(valid && bit);
      eq &= bit ? bm : ~bm;
    }
    const int clash = ballot64(valid && (eq & ~((u64)1 << lane)) != 0) != 0 ? 1 : 0;
    const bool want = valid && !((w >> (a & 31)) & 1u);
    const float dd = wave_distances_own<METRIC>(ix, a, want, qv, row_off);
    vb->row[sl][lane] = a;
    vb->loc[sl][lane] = loc;
    vb->dist[sl][lane] = dd;
    const u64 wm = ballot64(want);
    asm volatile("" ::"v"(touch));  // (the touch is a real load)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) {
      vb->mask[sl] = wm;
      vb->flags[sl] = clash;
      vb->tag[sl] = ((u64)(uint32_t)gen << 32) | (uint32_t)pnode;
    }
  }
}

// Distances of a row's own entries: every lane flagged `take` receives the distance of its entry `a`.  The flagged
// entries are packed onto lane pairs through the cross-lane network (ds_permute: the r-th flagged lane pushes its id to
// the pair r mod 32), a pair scores ONE row per pass (passes of 32; the second pass only while more than 32 entries
// are new, i.e. in a search's first hops), and the owner pulls the result back (ds_bpermute).  No trip of ids or
// results through LDS memory (wave_distances costs four LDS round trips per hop) and half the registers of the two-rows-
// per-pair routine; same arithmetic, same evaluation order.
// Both lanes of a pair receive the even lane's value.  The DPP move is pinned (empty asm) in the uniform block it is
// written in: hipcc otherwise turns `odd ? swap(v) : v` into a branch and runs the move with the even lanes disabled --
// and a DPP read of a disabled lane returns 0.
__device__ __forceinline__ int pair_even_value(int v) {
  int sw = __builtin_amdgcn_update_dpp(0, v, 0xB1 /* quad_perm:[1,0,3,2] */, 0xF, 0xF, true);
  asm volatile("" : "+v"(sw));
  return (lane_id() & 1) ? sw : v;
}

// LEAN: the routines that keep half a row per lane pair in flight (same arithmetic, same order; for callers short of registers).
template <int METRIC, bool LEAN>
__device__ __forceinline__ float wave_distances_own(const IndexView &ix, int a, bool take, const float *qv, int64_t row_off) {
  const int lane = lane_id();
  const int h = lane & 1;
  const u64 tm = ballot64(take);
  const int nt = popc64(tm);
  const int r = popc64(tm & lanemask_lt());
  float mine = 0.f;
  for (int base = 0; base < nt; base += 32) {
    // the r-th flagged lane -> even lane of pair r - base (the others push to lane 1, which nobody reads)
    const bool now = take && r >= base && r < base + 32;
    const int got = __builtin_amdgcn_ds_permute(now ? ((r - base) << 3) : 4, a);
    const int ev = pair_even_value(got);  // both lanes of the pair
    const int s = lane >> 1;
    const int id = (base + s < nt) ? ev : 0;  // idle pairs score node 0: no branches
    const float *prow = ix.points + (row_off + id) * (int64_t)ix.stride;
    float dd;
#if WANN_DT != 0
    dd = byte_pair<METRIC>(prow, qv, ix.stride, h);
#else
    if (LEAN) {
      if (METRIC == 1) dd = mips_pair(prow, qv, ix.d, h);
      else dd = l2_pair<8>(prow, qv, (ix.d + 7) >> 3, h, true);
    } else if (METRIC == 1) {
      const int np = (((ix.d + 3) >> 2) + 1) >> 1;
      switch (np) {  // wave-uniform
        case 12: dd = mips_pair_ct<12>(prow, qv, ix.d, h); break;
        case 13: dd = mips_pair_ct<13>(prow, qv, ix.d, h); break;
        default: dd = mips_pair(prow, qv, ix.d, h); break;
      }
    } else {
      const int D8 = (ix.d + 7) >> 3;
      switch (D8) {  // wave-uniform
        case 16: dd = l2_pair_ct<16>(prow, qv, h, true); break;
        case 12: dd = l2_pair_ct<12>(prow, qv, h, true); break;
        case 13: dd = l2_pair_ct<13>(prow, qv, h, true); break;
        case 8: dd = l2_pair_ct<8>(prow, qv, h, true); break;
        default: dd = l2_pair<16>(prow, qv, D8, h, true); break;
      }
    }
#endif
    // the result sits in the odd lane of the pair (in both for the inner product): the owner pulls it
    const float back = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((((r - base) << 1) | 1) << 2, __builtin_bit_cast(int, dd)));
    if (now) mine = back;
  }
  return mine;
}

template <int METRIC>
__device__ __forceinline__ void wave_beam_search_big(const IndexView &ix, const PartDesc &part, const WaveLds &L,
                                                     int32_t *gtable, uint32_t tag, uint32_t *gseen, int B, int bits,
                                                     int64_t qid, int64_t limit, int degree_limit, int32_t *mini,
                                                     uint32_t mini_mask, int &m_out, long long &nvis_out,
                                                     long long &ncmp_out, unsigned long long *prof = nullptr,
                                                     ScoreBox *box = nullptr, int part_index = 0,
                                                     const int32_t *abort_flag = nullptr, Counters *ctr = nullptr,
                                                     const long long *moot_word = nullptr, int my_level = 0) {
  // abort_flag: a look-ahead search (k_search) that its chain has withdrawn (*abort_flag == 2) stops at the next check
  // moot_word: a speculated level stops when a LOWER level of its task has found k entries (*moot_word < my_level)
  //
  // A search wave runs alone on its SIMD, so every dependent instruction costs its full latency and every trip between
  // the vector and the scalar side (ballot -> branch -> readlane ...) a few dozen cycles.  The hop below therefore keeps
  // its working set in registers: a 64-entry WINDOW of the LDS beam around the first unvisited entry with its unvisited
  // mask (the next node is a bit scan, not a beam read), the delta list's first unvisited and last keys and the cutoff as
  // scalars, the beam's last 64 entries as a TAIL cache for the truncation, and packet requests in beam order (the k-th
  // outstanding request belongs to the k-th unvisited entry: finding a packet needs no lookup).
  prof = WANN_PROF_PTR(prof);
  // Every wave-uniform argument into scalar registers, explicitly: ONE value the compiler can not prove uniform (a pointer
  // selected by a loaded flag, say) in ONE exit test makes the whole hop loop "divergent" -- and then every loop-carried
  // scalar (list sizes, window mask, cutoff ...) lives in vector registers under exec masks.
  const bool check_abort = uni((int)(abort_flag != nullptr || moot_word != nullptr)) != 0;
  my_level = uni(my_level);
  tag = (uint32_t)uni((int)tag);
  B = uni(B);
  bits = uni(bits);
  qid = uni64(qid);
  limit = uni64(limit);
  degree_limit = uni(degree_limit);
  mini_mask = (uint32_t)uni((int)mini_mask);
  part_index = uni(part_index);
  const int lane = lane_id();
  const uint32_t tmask = (1u << bits) - 1u;
  const int64_t row_off = uni(part.start);
  const int64_t row_base = uni64(part.row_base);
  const int part_n = uni(part.n);
  u64 *const mb = L.lbeam;
  LdsBox *const vb = lds_box(box);
  int my_gen = 0, st_pk = 0, st_own = 0, st_nx = 0;
  if (box) {  // a new search for the helper waves (the query is staged already)
    if (lane == 0) {
      vb->part = part_index;
      vb->bits = bits;
      vb->qid_lo = (int32_t)(uint32_t)qid;
      vb->qid_hi = (int32_t)(qid >> 32);
      vb->req_head = 0;
      my_gen = vb->last_gen + 1;
      vb->last_gen = my_gen;
    }
    my_gen = uni(my_gen);
  }
  {  // exact seen set of this search: empty
    int4 *sv = reinterpret_cast<int4 *>(gseen);
    const int n16 = (part_n + 127) >> 7;
    for (int i = lane; i < n16; i += 64) sv[i] = make_int4(0, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // in L2 before the first probe (probes and updates are L2 atomics)
  }
  // frontier = {start node 0} (beamSearch.h:80-82); the start node counts as scored
  if (lane == 0) L.cand_id[0] = 0;
  WAVE_SYNC();
  float d0 = wave_distances<METRIC>(ix, L.cand_id, L.cand_dist, L.qv, 1, row_off);
  d0 = __shfl(d0, 0);
  const u64 key0 = (u64)fkey(d0) << 32;
  if (lane == 0) {
    mb[0] = key0;
    __hip_atomic_fetch_or(gseen, 1u, __ATOMIC_RELAXED, kSeenScope);
    if (box) vb->gen = my_gen;  // (after the seen set's clear: the helpers read it)
  }
  WAVE_SYNC();
  int M = 1, D = 0;               // entries in the LDS beam / in the delta list
  u64 dk = ~0ull;                 // delta list: lane i < D holds its i-th smallest key; ~0 elsewhere
  u64 dhead = ~0ull, dlast = 0;   // its smallest unvisited key (~0: none) and its last key (0: empty), wave-uniform
  u64 mlk = key0;                 // key of the last entry of the LDS beam, mb[M - 1] (0: empty)
  float cutoff = 2147483648.0f;   // (float)INT_MAX while the beam is not full, else its last distance (beamSearch.h:135-137)
  // window: wv = mb[wbase + lane]; bit i of wum: entry wbase + i is unvisited (and exists); pmk = the first such key (~0: none)
  int wbase = 0;
  u64 wv = lane == 0 ? key0 : 1ull, wum = 1ull, pmk = key0;
  int tb = 0;                     // tail cache: tv = mb[tb + lane]
  u64 tv = lane == 0 ? key0 : 0ull;
  // packet requests: entries of the LDS beam up to position req_pos have been requested, in beam order; rq_next is the
  // request of the first unvisited one among them (requests rq_next .. nreq - 1 are outstanding)
  int nreq = 0, rq_next = 0, req_pos = -1;
  // the node this wave expects to visit next, with its packet and its filter / seen-set probes (issued during this hop's
  // insertion, consumed by the next hop if the expectation holds)
  int nx_node = -1, nx_a = -1, nx_old = -1, nx_flags = 0;
  uint32_t nx_loc = 0, nx_sw = 0;
  float nx_dist = 0.f;
  u64 nx_mask = 0;
  int nvis = 0, ncmp_v = 0;  // (dist_cmps: counted per lane, summed at the end)
  const int lim = limit > 0x7fffffff ? 0x7fffffff : (int)limit;

  auto load_window = [&](int s) {  // the first chunk at or after s that holds an unvisited entry
    for (;;) {
      const int x = s + lane;
      wv = x < M ? mb[x] : 1ull;
      wum = ballot64(!(wv & 1ull));
      wbase = s;
      if (wum || s + 64 >= M) break;
      s += 64;
    }
    pmk = wum ? rdlane64(wv, ctz64(wum)) : ~0ull;
  };
  auto load_tail = [&]() {
    tb = M > 64 ? M - 64 : 0;
    const int x = tb + lane;
    tv = x < M ? mb[x] : 0ull;
    mlk = M ? rdlane64(tv, M - 1 - tb) : 0ull;
  };
  auto set_cutoff = [&]() {
    cutoff = 2147483648.0f;
    if (M + D >= B) cutoff = funkey((uint32_t)(((mlk | 1ull) > (dlast | 1ull) ? mlk : dlast) >> 32));
  };
  auto forget_requests = [&]() {  // nothing at or after the first unvisited entry counts as requested
    rq_next = nreq;
    req_pos = wbase + (wum ? ctz64(wum) : 64) - 1;
  };
  // everything above from (mb, M, dk, D) -- after a merge of the LDS beam
  auto resync = [&](int s) {
    load_tail();
    load_window(s);
    const u64 du = ballot64(lane < D && !(dk & 1ull));
    dhead = du ? rdlane64(dk, ctz64(du)) : ~0ull;
    dlast = D ? rdlane64(dk, D - 1) : 0ull;
    set_cutoff();
    forget_requests();
    nx_node = -1;
  };
  // the complete packet of `node` in slot sl, if its helper has finished it: every field is read between two reads of the
  // slot's tag (the helper zeroes the tag before it rewrites a slot; LDS operations of a wave execute in order)
  auto read_packet = [&](int sl, int node, int &pa, uint32_t &ploc, float &pdist, u64 &pmask, int &pflags) -> bool {
    const u64 want = ((u64)(uint32_t)my_gen << 32) | (uint32_t)node;
    const u64 t1 = vb->tag[sl];
    const int ra = vb->row[sl][lane];
    const uint32_t rl = vb->loc[sl][lane];
    const float rd = vb->dist[sl][lane];
    const u64 rm = vb->mask[sl];
    const int rf = vb->flags[sl];
    const u64 t2 = vb->tag[sl];
    if (!uni((int)(t1 == want && t2 == want))) return false;
    pa = ra;
    ploc = rl;
    pdist = rd;
    pmask = rm;         // (the same value in every lane)
    pflags = rf | 2;    // bit 1: the row's slot-sharing test is known (bit 0)
    return true;
  };
  // (the flag words of wave_merge_run live in the clash test's scratch beside the beam -- the two are never in use together)
  const bool run_merge = (int)mini_mask + 1 >= ((B + 64 + 127) >> 5) + 2 && mini != reinterpret_cast<int32_t *>(L.cand_key);
  auto flush = [&]() {  // merge the delta list into the LDS beam
    if (D == 0) return;
    const int pm = wum ? wbase + ctz64(wum) : M;
    int p0;
    if (WANN_LIKELY(run_merge)) M = wave_merge_run(mb, M, dk, D, mini, &p0);
    else M = wave_merge<u64 *, false, true>(mb, M, B, lane < D, dk, L.cand_key, &p0);
    D = 0;
    dk = ~0ull;
    resync(pm < p0 ? pm : p0);
  };

  // ---- Into the delta list: the keys of the lanes flagged in pm (c of them, D + c <= 64; a computed candidate is always new,
  //      so all keys are distinct).  ONE pass over the candidates gives every list entry the number of candidates below it
  //      and every candidate the number of list entries and of other candidates below it -- i.e. everybody's place --, one trip
  //      through the merge scratch moves everything (a DPP shift of the whole list per candidate before: 45 instructions
  //      each, this pass: 14).
  auto delta_insert = [&](u64 pm, u64 key, bool pass) {
    const int c = popc64(pm);
    const u64 dk1 = dk | 1ull, key1 = key | 1ull;
    int shift = 0, below = 0, rank = 0;
    for (u64 mm = pm; mm; mm &= mm - 1) {
      const int i = ctz64(mm);
      const u64 k1 = rdlane64(key, i) | 1ull;
      const bool gt = dk1 > k1;  // (empty lanes hold ~0: true)
      shift += gt ? 1 : 0;
      const int less = 64 - popc64(ballot64(gt));  // list entries below candidate i
      below = (lane == i) ? less : below;
      rank += (key1 > k1) ? 1 : 0;
    }
    if (lane < D) L.cand_key[lane + shift] = dk;
    if (pass) L.cand_key[below + rank] = key;
    WAVE_SYNC();
    D += c;
    dk = lane < D ? L.cand_key[lane] : ~0ull;
    WAVE_SYNC();
    const u64 du = ballot64(!(dk & 1ull));  // (empty lanes hold ~0: bit 0 set)
    dhead = du ? rdlane64(dk, ctz64(du)) : ~0ull;
    dlast = rdlane64(dk, D - 1);
  };
  // ---- Truncate to B (beamSearch.h:157): the `excess` largest entries of (LDS beam, delta list) leave; both are sorted
  auto truncate = [&]() {
    int excess = M + D - B;
    if (excess <= 0) return;
    // the usual case in one step: everything that leaves is in the LDS beam (its entry M - excess is beyond the delta list's last)
    const int q = M - excess;
    if (WANN_LIKELY(q >= 1 && q - 1 >= tb) && (rdlane64(tv, q - tb) | 1ull) > (dlast | 1ull)) {
      M = q;
      mlk = rdlane64(tv, M - 1 - tb);
      return;
    }
    for (; excess > 0; excess--) {
      if (D > 0 && (M == 0 || (dlast | 1ull) > (mlk | 1ull))) {
        if (lane == D - 1) dk = ~0ull;
        D--;
        if (dlast == dhead) dhead = ~0ull;  // (it was the only unvisited entry)
        dlast = D ? rdlane64(dk, D - 1) : 0ull;
      } else {
        M--;
        if (M == 0) mlk = 0;
        else if (WANN_LIKELY(M - 1 >= tb)) mlk = rdlane64(tv, M - 1 - tb);
        else load_tail();
      }
    }
  };
  auto window_lost = [&]() {  // after a truncation: the window may have lost entries
    if (WANN_UNLIKELY(M < wbase + 64)) {
      const int keep = M - wbase;
      wum = keep <= 0 ? 0ull : (wum & ((((u64)1 << (keep - 1)) << 1) - 1));
      if (!wum) pmk = ~0ull;
      if (M - 1 < req_pos) {
        forget_requests();
        nx_node = -1;
      }
    }
  };
  // packets are requested for the next WANN_BIG_LOOKAHEAD unvisited entries of the window
  auto request_packets = [&]() {
#pragma unroll
    for (int r = 0; r < 2; r++) {  // (one per hop keeps the distance; two catch up after a restart)
      if (r == 1 && WANN_LIKELY(nreq - rq_next >= WANN_BIG_LOOKAHEAD)) break;
      u64 cand = wum;
      const int rel = req_pos - wbase;  // window bits <= rel have been requested
      if (rel >= 63) cand = 0;
      else if (rel >= 0) cand &= ~(((u64)2 << rel) - 1);
      if (!cand) break;
      const int j = ctz64(cand);
      if (popc64(wum & (((u64)1 << j) - 1)) >= WANN_BIG_LOOKAHEAD) break;
      const int node = (int)((uint32_t)rdlane((int)(uint32_t)wv, j) >> 1);
      if (lane == 0) {
        vb->req[nreq & (kReqRing - 1)] = node;
        vb->req_head = nreq + 1;
      }
      nreq++;
      req_pos = wbase + j;
    }
  };
  // the next hop's node, as far as one can tell now (a candidate of this hop may still come first): its packet and its probes
  // are fetched while this hop's candidates are inserted.  The probes are issued after this hop's filter stores and seen-set
  // updates (same wave, program order), so they see them.
  auto fetch_next = [&]() {
    if (box && wum && !((dhead | 1ull) < (pmk | 1ull)) && wbase + ctz64(wum) <= req_pos) {
      const int nn = (int)((uint32_t)pmk >> 1);
      if (read_packet(rq_next % kPkSlots, nn, nx_a, nx_loc, nx_dist, nx_mask, nx_flags)) {
        nx_node = nn;
        // (no lane-dependent branch around the loads: an unused slot's filter slot lies inside the table like any other,
        // and its seen-set word is read at node 0)
        nx_old = gtable[nx_loc & tmask];
        nx_sw = __hip_atomic_load(gseen + ((nx_a < 0 ? 0 : nx_a) >> 5), __ATOMIC_RELAXED, kSeenScope);
      }
    }
  };

  unsigned long long tp = 0, acc[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // (11 .. 13: hops from the delta list / window hops without a request / with a request but no packet yet)
#define WANN_PHASE(i)                                       \
  do {                                                      \
    if (prof) {                                             \
      unsigned long long tn = __builtin_readcyclecounter(); \
      acc[i] += tn - tp;                                    \
      tp = tn;                                              \
    }                                                       \
  } while (0)
  if (prof) tp = __builtin_readcyclecounter();
  for (;;) {
    // The loop-carried scalars, declared uniform once per hop.  For the compiler a value is divergent as soon as it is merged
    // at the join of ANY lane-dependent branch (`if (valid) store` next to `flag = ...` is enough), and one such value in
    // one exit test turns the whole loop into an exec-masked one with every scalar in vector registers.  A readfirstlane of
    // a value the compiler already knows to be uniform folds away.
#define WANN_UNIFORM_STATE()                                              \
  do {                                                                    \
    M = uni(M);                                                           \
    D = uni(D);                                                           \
    wbase = uni(wbase);                                                   \
    tb = uni(tb);                                                         \
    nreq = uni(nreq);                                                     \
    rq_next = uni(rq_next);                                               \
    req_pos = uni(req_pos);                                               \
    nx_node = uni(nx_node);                                               \
    nvis = uni(nvis);                                                     \
    wum = (u64)uni64((long long)wum);                                     \
    pmk = (u64)uni64((long long)pmk);                                     \
    dhead = (u64)uni64((long long)dhead);                                 \
    dlast = (u64)uni64((long long)dlast);                                 \
    mlk = (u64)uni64((long long)mlk);                                     \
    cutoff = __builtin_bit_cast(float, uni(__builtin_bit_cast(int, cutoff))); \
  } while (0)
    WANN_UNIFORM_STATE();

    // ---- THE FAST PATH, a tight loop of its own: the hop whose node is the one the previous hop expected (the first unvisited
    //      entry of the window, not of the delta list), with its packet and both probes fetched already, in a row without
    //      shared filter slots.  Nine hops in ten of a long search.  Same steps, same order as the general hop below -- minus
    //      every branch that hop needs for the rest, so that the compiler lays this one out straight and keeps its state in
    //      registers (a lone wave pays ~7 cycles per instruction: instructions are what a hop costs).
    while (WANN_LIKELY(box != nullptr && nx_node >= 0 && wum != 0 && nx_node == (int)((uint32_t)pmk >> 1) && !(uni(nx_flags) & 1) &&
                       !((dhead | 1ull) < (pmk | 1ull)) && nvis < lim && !(check_abort && (nvis & 31) == 0))) {
      const int i = ctz64(wum);
      if (lane == 0) mb[wbase + i] = pmk | 1ull;
      wum &= wum - 1;
      const bool consumed = wbase + i <= req_pos;
      nvis++;
      st_nx += (lane == 0) ? 1 : 0;
      st_pk += (lane == 0) ? 1 : 0;
      const int a = nx_a;
      const uint32_t loc = nx_loc, sw = nx_sw;
      const int old = nx_old;
      float dist = nx_dist;
      const u64 pk_mask = nx_mask;
      const bool valid = (a >= 0) && (lane < degree_limit) && ((int64_t)a != qid);
      if (consumed) rq_next++;
      nx_node = -1;
      WANN_PHASE(0);
      if (prof) {  // (profile builds: how long the probes issued during the previous hop are still in flight)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        WANN_PHASE(9);
      }
      if (WANN_UNLIKELY(popc64(wum) < 4 && wbase + 64 < M)) load_window(wum ? wbase + ctz64(wum) : wbase + 64);
      else pmk = wum ? rdlane64(wv, ctz64(wum)) : ~0ull;
      request_packets();
      WANN_PHASE(2);
      // lossy seen-filter (no shared slots in this row: the sequential rule is "old == id") + exact seen set
      // (The compiler would issue the filter's store BEFORE it waits for the second probe; that wait then also covers the store's
      // round trip -- the vector-memory counter retires in order --, i.e. two dependent round trips per hop: store, then the next
      // hop's probes.  The empty asm below makes both probes "used" before the store.  Alone that is SLOWER -- the search wave
      // outruns its helpers: 16 % of the hops without a packet instead of 7 % -- and with requests eight entries ahead instead of
      // four it is faster: see WANN_BIG_LOOKAHEAD.)
      const int tagged = (int)(tag | (uint32_t)a);
      const bool seen = valid && (old == tagged);
      const bool kept = valid && !seen;  // what the reference scores
      ncmp_v += kept ? 1 : 0;
      const bool take = kept && !((sw >> (a & 31)) & 1u);  // what is computed
      asm volatile("" ::"v"(sw), "v"(old));
      if (kept) gtable[loc] = tagged;  // (a slot that holds the id already is left alone: its line -- of a table of up to 32 MiB -- stays clean)
      if (take) __hip_atomic_fetch_or(gseen + (a >> 5), 1u << (a & 31), __ATOMIC_RELAXED, kSeenScope);
      WANN_PHASE(4);
      fetch_next();
      WANN_PHASE(5);
      const bool need = take && !((pk_mask >> lane) & 1ull);
      if (WANN_UNLIKELY(ballot64(need) != 0)) {
        st_own += (lane == 0) ? 1 : 0;
        L.cand_key[lane] = dk;
        const float own = wave_distances_own<METRIC, true>(ix, a, need, L.qv, row_off);
        dk = L.cand_key[lane];
        if (need) dist = own;
      }
      const bool pass = take && (dist < cutoff);
      const u64 key = ((u64)fkey(dist) << 32) | ((u64)(uint32_t)a << 1);
      const u64 pmask = ballot64(pass);
      WANN_PHASE(6);
      if (pmask) {
        if (WANN_UNLIKELY(D + popc64(pmask) > 64)) {
          flush();
          WANN_PHASE(10);
        }
        delta_insert(pmask, key, pass);
        WANN_PHASE(7);
        truncate();
        window_lost();
        set_cutoff();
        WANN_PHASE(8);
      }
      WANN_UNIFORM_STATE();
    }

    // ---- visit the closest unvisited entry of the beam (beamSearch.h:108-117): the closer of the first unvisited
    //      entry of the LDS beam and the first unvisited entry of the delta list
    if ((pmk == ~0ull && dhead == ~0ull) || nvis >= lim) break;
    if (WANN_UNLIKELY(check_abort && (nvis & 31) == 0)) {
      int ab = 0;
      if (lane == 0) {
        if (abort_flag) ab = __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (moot_word && __hip_atomic_load(moot_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < my_level) ab = 2;
      }
      if (uni(ab) == 2) break;
    }
    const bool from_delta = (dhead | 1ull) < (pmk | 1ull);
    int cur;
    bool consumed = false;  // the visited entry had a packet requested
    if (from_delta) {
      cur = (int)((uint32_t)dhead >> 1);
      if (dk == dhead) dk |= 1ull;  // (keys are unique)
      const u64 du = ballot64(lane < D && !(dk & 1ull));
      dhead = du ? rdlane64(dk, ctz64(du)) : ~0ull;
    } else {
      const int i = ctz64(wum);
      cur = (int)((uint32_t)pmk >> 1);
      if (lane == 0) mb[wbase + i] = pmk | 1ull;
      wum &= wum - 1;
      consumed = wbase + i <= req_pos;
    }
    nvis++;
    WANN_PHASE(0);  // select
    if (prof) {  // (profile builds: how long the probes issued during the previous hop are still in flight)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      WANN_PHASE(9);
    }

    // ---- adjacency row (graph.h:198; -1 = unused slot), the neighbours' filter slots and the two probes: prepared during
    //      the previous hop if this is the node it expected; else from a helper's packet; else from memory
    int a = -1, old = -1, flags = 0;
    uint32_t loc = 0, sw = 0;
    float pk_dist = 0.f;
    u64 pk_mask = 0;
    bool valid;
    if (cur == nx_node && !from_delta) {
      st_nx += (lane == 0) ? 1 : 0;
      a = nx_a;
      loc = nx_loc;
      old = nx_old;
      sw = nx_sw;
      pk_dist = nx_dist;
      pk_mask = nx_mask;
      flags = nx_flags;
      valid = (a >= 0) && (lane < degree_limit) && ((int64_t)a != qid);
    } else {
      bool got = false;
      if (box && consumed) got = read_packet(rq_next % kPkSlots, cur, a, loc, pk_dist, pk_mask, flags);
      else if (box && !from_delta && nreq > 0) {
        // An entry of the LDS beam whose request does not count any more: every merge of the delta list into the beam forgets the
        // outstanding requests (new entries may sit between them), and the first hops after it found no packet -- one hop in
        // twenty of a long search, each three dependent round trips of this wave (row, probes, vectors); half of them are
        // found this way (2^-9 batch: hops without a packet 8.2 -> 6.0 %, 11.23 -> 10.98 ms).  The packets are
        // still in their slots, though, tagged with their node: look the node up in the request ring (lane r reads ring
        // entry r; entry r holds request idx, the last one issued with idx % kReqRing == r) and take the most recent hit
        // whose packet is complete.
        const int rn = lane < kReqRing ? vb->req[lane] : -1;
        const int idx = nreq - 1 - ((nreq - 1 - lane) & (kReqRing - 1));
        u64 hit = ballot64(lane < kReqRing && idx >= 0 && rn == cur);
        while (hit && !got) {
          int best = -1;
          for (u64 hm = hit; hm; hm &= hm - 1) {
            const int v = rdlane(idx, ctz64(hm));
            best = v > best ? v : best;
          }
          got = read_packet(best % kPkSlots, cur, a, loc, pk_dist, pk_mask, flags);
          hit &= ~((u64)1 << (best & (kReqRing - 1)));
        }
      }
      if (!got) {
        if (lane < ix.rs) a = ix.graph[(row_base + cur) * (int64_t)ix.rs + lane];
        loc = (uint32_t)hash64_2((u64)(uint32_t)a) & tmask;
      }
      valid = (a >= 0) && (lane < degree_limit) && ((int64_t)a != qid);
      if (valid) {
        old = gtable[loc];
        sw = __hip_atomic_load(gseen + (a >> 5), __ATOMIC_RELAXED, kSeenScope);
      }
    }
    if (consumed) rq_next++;
    nx_node = -1;
    st_pk += (lane == 0 && (flags & 2)) ? 1 : 0;  // (lane-dependent on purpose: a statistic must not cost a scalar register)
    if (prof) {
      acc[11] += from_delta ? 1 : 0;
      acc[12] += (!from_delta && !consumed) ? 1 : 0;
      acc[13] += (!from_delta && consumed && !(flags & 2)) ? 1 : 0;
    }
    WANN_PHASE(1);  // row, filter slots, probes

    // ---- the window follows the first unvisited entry
    if (!from_delta) {
      if (WANN_UNLIKELY(popc64(wum) < 4 && wbase + 64 < M)) load_window(wum ? wbase + ctz64(wum) : wbase + 64);
      else pmk = wum ? rdlane64(wv, ctz64(wum)) : ~0ull;
    }
    if (box) request_packets();
    WANN_PHASE(2);  // next unvisited entry + packet requests

    // ---- lossy seen-filter (sequential semantics, beamSearch.h:68-73,126-131) + exact seen set
    const int tagged = (int)(tag | (uint32_t)a);
    bool clash;
    if (flags & 2) clash = (flags & 1) != 0;  // (the helper's exact test)
    else {
      // exact test "two valid lanes of the row share a filter slot": every lane tags its slot of a small LDS hash with
      // its lane number; a lane that lost its slot compares filter slots with the winner, and the few lanes whose
      // loss was a collision of the small hash only are compared with all lanes
      const uint32_t mh = loc & mini_mask;
      if (valid) mini[mh] = lane;
      WAVE_SYNC();
      const int mw = valid ? mini[mh] : lane;
      WAVE_SYNC();
      const uint32_t loc_w = (uint32_t)__shfl((int)loc, mw);
      const bool lost = valid && (mw != lane);
      clash = ballot64(lost && loc_w == loc) != 0;
      for (u64 um = ballot64(lost && loc_w != loc); um && !clash; um &= um - 1) {
        const int u = ctz64(um);
        const uint32_t lu = (uint32_t)rdlane((int)loc, u);
        clash = ballot64(valid && loc == lu && lane != u) != 0;
      }
    }
    clash = uni((int)clash) != 0;
    WANN_PHASE(3);  // slot-sharing test
    bool seen, twice = false;
    if (WANN_LIKELY(!clash)) {
      seen = valid && (old == tagged);
      if (valid && !seen) gtable[loc] = tagged;
    } else {  // exact emulation of the sequential rule (as in wave_beam_search)
      u64 eq = ballot64(valid);
      for (int b = 0; b < bits; b++) {
        const bool bit = (loc >> b) & 1u;
        const u64 bm = ballot64(valid && bit);
        eq &= bit ? bm : ~bm;
      }
      const u64 lower = valid ? (eq & lanemask_lt()) : 0ull;
      const u64 higher = (lane == 63) ? 0ull : (eq >> (lane + 1));
      const int prev_lane = lower ? (63 - __builtin_clzll(lower)) : lane;
      const int prev_val = __shfl(a, prev_lane);
      seen = valid && (lower ? (prev_val == a) : (old == tagged));
      WAVE_SYNC();
      if (valid && higher == 0) gtable[loc] = tagged;
      // does the row list one node twice (the reference's builder can append the start point twice)?
      u64 lm = lower;
      while (ballot64(lm != 0)) {
        const int l = lm ? (63 - __builtin_clzll(lm)) : lane;
        const int v = __shfl(a, l);
        if (lm && v == a) twice = true;
        if (lm) lm &= ~((u64)1 << l);
      }
      twice = ballot64(twice) != 0;
    }
    twice = uni((int)twice) != 0;  // (a uniform flag merged at the join of a lane-dependent branch counts as divergent)
    const bool kept = valid && !seen;  // what the reference scores
    ncmp_v += kept ? 1 : 0;
    const bool fresh = kept && !((sw >> (a & 31)) & 1u);
    const bool take = twice ? kept : fresh;  // what is computed
    if (take) __hip_atomic_fetch_or(gseen + (a >> 5), 1u << (a & 31), __ATOMIC_RELAXED, kSeenScope);
    WANN_PHASE(4);  // seen-filter

    fetch_next();
    WANN_PHASE(5);  // next hop's packet and probes

    if (WANN_UNLIKELY(twice)) flush();  // the exact multiset union below works on the whole beam
    // ---- score (beamSearch.h:135-145).  Distances: from the packet where it holds them; what it lacks (no packet, or a
    //      neighbour the helper took for scored already) is computed here
    float dist = pk_dist;
    const bool need = take && !((pk_mask >> lane) & 1ull);
    if (ballot64(need) != 0) {
      st_own += (lane == 0) ? 1 : 0;
      // (the delta list waits in the merge scratch meanwhile: the scoring routine keeps a whole row per lane pair in
      // flight and needs every register)
      L.cand_key[lane] = dk;
      const float own = wave_distances_own<METRIC, true>(ix, a, need, L.qv, row_off);
      dk = L.cand_key[lane];
      if (need) dist = own;
    }
    const bool pass = take && (dist < cutoff);
    const u64 key = ((u64)fkey(dist) << 32) | ((u64)(uint32_t)a << 1);
    const u64 pmask = ballot64(pass);
    WANN_PHASE(6);  // distances

    // ---- union + truncate (beamSearch.h:148-157)
    if (WANN_UNLIKELY(twice)) {
      int p0;
      const int pm = wum ? wbase + ctz64(wum) : M;
      M = wave_merge(mb, M, B, pass, key, L.cand_key, &p0);
      resync(pm < p0 ? pm : p0);
    } else if (pmask) {
      if (WANN_UNLIKELY(D + popc64(pmask) > 64)) {
        flush();
        WANN_PHASE(10);
      }
      delta_insert(pmask, key, pass);
      WANN_PHASE(7);  // into the delta list
      truncate();
      window_lost();
      set_cutoff();
    }
    WANN_PHASE(8);  // truncation
  }
#undef WANN_UNIFORM_STATE
#undef WANN_PHASE
  if (box && lane == 0) vb->gen = 0;  // (the next search picks the next generation)
  if (D) {
    int p0;
    if (run_merge) M = wave_merge_run(mb, M, dk, D, mini, &p0);
    else M = wave_merge<u64 *, false, true>(mb, M, B, lane < D, dk, L.cand_key, &p0);
  }
  for (int o = 32; o; o >>= 1) ncmp_v += __shfl_xor(ncmp_v, o);
  if (prof && lane == 0)
    for (int i = 0; i < 14; i++) atomicAdd(&prof[i], acc[i]);
  if (ctr && lane == 0) {
    atomicAdd(&ctr->big_searches, 1ull);
    atomicAdd(&ctr->big_hops, (unsigned long long)nvis);
    atomicAdd(&ctr->packet_hops, (unsigned long long)st_pk);
    atomicAdd(&ctr->own_scorings, (unsigned long long)st_own);
    atomicAdd(&ctr->prefetched_hops, (unsigned long long)st_nx);
  }
  m_out = M;
  nvis_out = nvis;
  ncmp_out = 1 + ncmp_v;
}

// --------------------------------------------------------------------------------------------
// Third-generation general core: beams that do not fit the register-resident variant, in the FOUR-wave kernel
// (129 .. its cap; round 5).  The search of wave_beam_search<.., false, true, false> -- sorted beam in the LDS, lossy
// seen-filter in global memory, std::set_union semantics -- restructured around what the counters showed: a hop of that core
// was ~800 instructions and ~25 DEPENDENT LDS round trips (the union's search / duplicate test / shift, the slot-sharing
// test, compaction) beside its three dependent memory round trips (adjacency row -> filter probes -> vectors), and under
// load it was bound by filter lines (a 128-byte line per probe and per store of a table of up to 2 MiB per search).
//
//  * EXPECTATIONS.  Which node the next hops visit is nearly always known: the first unvisited beam entries behind the
//    current one (measured on the oracle: 89 % of the hops at beam 160, 94 % at 320, 97 % at 640, 98.5 % at 1 280).  The rows
//    of the next TWO expected nodes are requested as soon as they are known (slots s1 / s2, a register each), and s1's filter
//    probes right after the current hop's filter stores (same wave, program order: they see them).  Row, filter slots and
//    slot-sharing test are pure functions of the node; the probes are valid exactly while no other hop's stores followed them.
//  * THE UNION IS DEFERRED.  Candidates that pass the cutoff go to a 64-entry PENDING buffer; the LDS beam is left alone.
//    The reference's beam is (LDS beam U pending), truncated to B.  The next hop visits s1 -- the first unvisited entry of the
//    LDS beam -- iff no pending key sorts at or before it (one scalar compare against the smallest pending key); then s1 is
//    the closest unvisited entry of the whole beam and its rank there is its position, i.e. below B.  Otherwise the pending
//    candidates are united first (ONE wave_merge for all of them) and the search goes on from the exact beam; so do the end of
//    the search and the hop limit.  Between unions the cutoff is the LDS beam's last distance: never below the reference's,
//    so a candidate the reference would have rejected may be admitted.  With a distance ABOVE the reference's B-th entry's it
//    sorts behind that entry, is never visited (rule above) and leaves at the next truncation.  With an EQUAL distance and a
//    smaller id it would sort before it and displace it (the reference rejects dist >= cutoff): a candidate at or above a
//    lower bound of the true cutoff (the LDS beam's entry at B - 1 - D) whose distance equals that of an entry that may be
//    the B-th one takes the exact path -- pending united first, test repeated.  Equal pending keys come from different hops (a hop without
//    shared filter slots lists every node once), where the reference's per-hop union keeps one copy: wave_merge<COLLAPSE>;
//    a hop WITH shared slots is united on its own, at once, with the multiset rule.
//  * A HOP'S REQUESTS GO OUT TOGETHER, the vectors first; expectations, probes and the pending buffer are worked on while
//    they travel.  The steady-state hop is one memory round trip + distance arithmetic + ~350 instructions.
//  * Vectors are requested at one point of the hop and consumed at another (RowRegs: half a row per lane, one row per
//    lane pair and pass, ids and results through the cross-lane network as in wave_distances_own) where the kernel's
//    register budget allows (elsewhere they are fetched where they are scored); a second pass's rows are TOUCHED at request
//    time (one dword per 128-byte line) so that their round trip ends in the L2.
//  * Filter entries are tagged with the slot's search epoch (no table clear per search), a slot that already holds the id is
//    not stored to, and a "slot written" bitmap in the LDS suppresses the probes of slots this search has not written.
//
// Everything with sequential semantics -- lossy filter, visit order, union -- follows the reference; results, hops and
// dist_cmps are those of wave_beam_search (the parity tests run every core against the oracle).
// --------------------------------------------------------------------------------------------
// (NR = the most blocks a lane holds: what the kernel's register budget affords -- 16 in the squared-L2 float kernel (two waves
// per SIMD), fewer or none in the kernels built for three)
#if WANN_DT != 0
constexpr int kRowRegs = 4;  // byte rows of up to 128 elements (8 cost the byte kernels, built for three waves per SIMD, a scratch segment)
#else
constexpr int kRowRegsL2 = 16, kRowRegsMips = 0;
#endif
template <int NR>
struct RowRegs {
  float4 v[NR > 0 ? NR : 1];
};
template <int METRIC>
struct RowRegsFor {
#if WANN_DT != 0
  static constexpr int NR = kRowRegs;
#else
  static constexpr int NR = METRIC == 1 ? kRowRegsMips : kRowRegsL2;
#endif
  typedef RowRegs<NR> type;
};

// how many 16-byte blocks of a row a lane holds in RowRegs (wave-uniform); 0: this row shape is fetched where it is scored.
// (ONE loader and one scoring routine per metric, blocks predicated by the count: a switch over compile-time routines made
// hipcc merge the cases' loads into one block with an address register pair per load -- 270 registers.)
template <int METRIC>
__device__ __forceinline__ int row_regs_blocks(const IndexView &ix) {
#if WANN_DT != 0
  const int chunks = ix.stride >> 3;  // 16-byte chunks per lane
  return chunks <= RowRegsFor<METRIC>::NR ? chunks : 0;
#else
  if (METRIC == 1) {
    const int np = (((ix.d + 3) >> 2) + 1) >> 1;
    return np <= RowRegsFor<METRIC>::NR ? np : 0;
  }
  const int D8 = (ix.d + 7) >> 3;  // (an odd block count starts with the LAST block, NSGDist.h:40-47: only 13 -- d = 100 -- is served)
  return (D8 <= RowRegsFor<METRIC>::NR && (!(D8 & 1) || D8 == 13)) ? D8 : 0;
#endif
}

// lane h of a pair holds the 16-byte blocks 2 j + h of its row, j < nblk -- the same addresses for float32 rows under either
// metric and for byte rows
// NBC > 0: the block count is a compile-time constant (no branch per block, the loads go out back to back and the scoring waits
// for them one by one); NBC = 0: `nblk` decides at run time
template <int NR, int NBC = 0>
__device__ __forceinline__ void row_regs_load(RowRegs<NR> &rr, const float *__restrict__ prow, int h, int nblk) {
#pragma unroll
  for (int j = 0; j < NR; j++)
    if (NBC > 0 ? j < NBC : j < nblk) rr.v[j] = *reinterpret_cast<const float4 *>(prow + 8 * j + 4 * h);
}

template <int METRIC, int NR, int NBC = 0>
__device__ __forceinline__ float row_regs_score(const RowRegs<NR> &rr, const float *qv, const IndexView &ix, int h, int nblk) {
  if (NR == 0) return 0.f;
  if (NBC > 0) nblk = NBC;
#if WANN_DT != 0
  int ab = 0, aa = 0, qq = 0;  // byte_pair's arithmetic (exact integer sums: any order)
#pragma unroll
  for (int j = 0; j < NR; j++)
    if (j < nblk) {
      const uint4 q = *reinterpret_cast<const uint4 *>(qv + 8 * j + 4 * h);
      const uint4 p = __builtin_bit_cast(uint4, rr.v[j]);
      ab = dot4_acc(p.x, q.x, ab);
      ab = dot4_acc(p.y, q.y, ab);
      ab = dot4_acc(p.z, q.z, ab);
      ab = dot4_acc(p.w, q.w, ab);
      if (METRIC == 0) {
        aa = dot4_acc(p.x, p.x, aa);
        aa = dot4_acc(p.y, p.y, aa);
        aa = dot4_acc(p.z, p.z, aa);
        aa = dot4_acc(p.w, p.w, aa);
        qq = dot4_acc(q.x, q.x, qq);
        qq = dot4_acc(q.y, q.y, qq);
        qq = dot4_acc(q.z, q.z, qq);
        qq = dot4_acc(q.w, q.w, qq);
      }
    }
  const int mine = METRIC == 0 ? (aa + qq - 2 * ab) : ab;
  const int both = mine + __shfl_xor(mine, 1);
  return METRIC == 0 ? (float)both : -(float)both;
#else
  if (METRIC == 1) {  // mips_pair_ct's arithmetic: one running scalar, products rounded then added in index order, fused tail
    const int tail_from = (ix.d & ~7) >> 3;
    float r = 0.f;
#pragma unroll
    for (int t = 0; t < NR; t++)
      if (t < nblk) {
        const float4 q = *reinterpret_cast<const float4 *>(qv + 8 * t + 4 * h);
        r = mips_step(r, rr.v[t], q, t >= tail_from);
      }
    return -r;
  }
  // l2_pair_ct's arithmetic: blocks in the reference's order (an odd count: last block first)
  f32x2 alo = {0.f, 0.f}, ahi = {0.f, 0.f};
  int nfull = nblk;
  if (NR > 12 && nblk == 13) {
    const float4 q = *reinterpret_cast<const float4 *>(qv + 8 * 12 + 4 * h);
    sq_acc(alo, ahi, rr.v[NR > 12 ? 12 : 0], q);
    nfull = 12;
  }
#pragma unroll
  for (int i = 0; i < NR; i++)
    if (i < nfull) {
      const float4 q = *reinterpret_cast<const float4 *>(qv + 8 * i + 4 * h);
      sq_acc(alo, ahi, rr.v[i], q);
    }
  const float s = ((alo.x + alo.y) + ahi.x) + ahi.y;
  const float other = __shfl_xor(s, 1);
  return (((other + alo.x) + alo.y) + ahi.x) + ahi.y;
#endif
}

// Request the rows of a hop's kept neighbours (lanes flagged `take`, entry `a`): the r-th flagged lane pushes its id to lane
// pair r (first pass: r < 32), the pair's lanes request half a row each.  r / nt: rank of this lane among the flagged, their number.
template <int METRIC, int NBC = 0>
__device__ __forceinline__ void mid_request_rows(const IndexView &ix, int a, bool take, int64_t row_off, int mode,
                                                 typename RowRegsFor<METRIC>::type &rr, int &r, int &nt, int &touch) {
  const int lane = lane_id();
  const u64 tm = ballot64(take);
  nt = popc64(tm);
  r = popc64(tm & lanemask_lt());
  if (nt == 0) return;
  if (RowRegsFor<METRIC>::NR == 0 || mode == 0) return;  // (rows fetched where they are scored: see below)
  // The second pass's rows (more than 32 kept neighbours: a search's first hops, every other hop at beams of 160) are fetched
  // when the first pass has been scored -- a round trip of their own.  Their cache lines are requested NOW (one dword per
  // 128-byte line, the value is never looked at), so that round trip ends in the L2; the requests retire with this hop's other
  // requests.  (The same for ALL rows of a kernel without the registers -- mode 0 -- was measured and dropped: the real loads
  // follow the touches within ~1 500 cycles, too soon to gain from them, and under load 40 % of the touched lines were
  // fetched twice: 6 % slower alone, 10 % under load on an inner-product graph, FETCH_SIZE of the deep-like leg 1.15 -> 1.26 x.)
  if (WANN_UNLIKELY(nt > 32)) {
    const int got = __builtin_amdgcn_ds_permute((take && r >= 32) ? ((r - 32) << 2) : (63 << 2), a);  // lane j: the (32 + j)-th kept id
    const int lpr = (ix.stride * 4 + 127) >> 7, total = (nt - 32) * lpr;
    for (int t0 = 0; t0 < total; t0 += 64) {
      const int t = t0 + lane, j = t / lpr;
      const int idj = __builtin_amdgcn_ds_bpermute((j & 63) << 2, got);
      if (t < total) touch |= *reinterpret_cast<const int *>(ix.points + (row_off + idj) * (int64_t)ix.stride + (t - j * lpr) * 32);
    }
  }
  const bool now = take && r < 32;
  const int got = __builtin_amdgcn_ds_permute(now ? (r << 3) : 4, a);
  const int ev = pair_even_value(got);
  const int id = ((lane >> 1) < nt) ? ev : 0;  // idle pairs fetch node 0: no branches
  row_regs_load<RowRegsFor<METRIC>::NR, NBC>(rr, ix.points + (row_off + id) * (int64_t)ix.stride, lane & 1, mode);
}

// ... and their distances: every flagged lane receives the distance of its entry
template <int METRIC, int NBC = 0>
__device__ __forceinline__ float mid_take_distances(const IndexView &ix, int a, bool take, int64_t row_off, const float *qv, int mode,
                                                    const typename RowRegsFor<METRIC>::type &rr, int r, int nt, int touch) {
  asm volatile("" ::"v"(touch));  // (the second pass's lines: real loads, retired with the first pass's vectors)
  if (nt == 0) return 0.f;
  // (no registers held across the hop: the compile-time routines where the row shape has one -- a whole row per lane pair in flight,
  // one round trip per pass)
  // (a kernel that holds RowRegs for other row shapes has no room for a second whole row: the lean routines there)
  if (RowRegsFor<METRIC>::NR == 0 || mode == 0) return wave_distances_own<METRIC, (RowRegsFor<METRIC>::NR != 0)>(ix, a, take, qv, row_off);
  const int lane = lane_id(), h = lane & 1;
  float mine = 0.f;
  {
    const float dd = row_regs_score<METRIC, RowRegsFor<METRIC>::NR, NBC>(rr, qv, ix, h, mode);
    const float back = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(((r << 1) | 1) << 2, __builtin_bit_cast(int, dd)));
    if (take && r < 32) mine = back;
  }
  if (WANN_UNLIKELY(nt > 32)) {  // (more than 32 new neighbours: a search's first hops)
    const bool now = take && r >= 32;
    const int got = __builtin_amdgcn_ds_permute(now ? ((r - 32) << 3) : 4, a);
    const int ev = pair_even_value(got);
    const int id = (32 + (lane >> 1) < nt) ? ev : 0;
    typename RowRegsFor<METRIC>::type r2;  // (registers of its own: `rr` has ONE definition per hop, or the compiler copies it around)
    row_regs_load<RowRegsFor<METRIC>::NR, NBC>(r2, ix.points + (row_off + id) * (int64_t)ix.stride, h, mode);
    const float dd = row_regs_score<METRIC, RowRegsFor<METRIC>::NR, NBC>(r2, qv, ix, h, mode);
    const float back = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((((r - 32) << 1) | 1) << 2, __builtin_bit_cast(int, dd)));
    if (now) mine = back;
  }
  return mine;
}

// NBC: the blocks of a row a lane holds as a compile-time constant (the caller has checked row_regs_blocks() == NBC), 0 = by
// the index (run time)
template <int METRIC, int NBC = 0>
__device__ __forceinline__ void wave_beam_search_mid(const IndexView &ix, const PartDesc &part, const WaveLds &L, int32_t *gtable,
                                                     uint32_t tag, int B, int bits, int64_t qid, int64_t limit, int degree_limit,
                                                     int32_t *mini, uint32_t mini_mask, int &m_out, long long &nvis_out,
                                                     long long &ncmp_out, unsigned long long *prof = nullptr, uint32_t *wbits = nullptr,
                                                     int wwords = 0, int wshift = 0) {
  // wbits (LDS, wwords 32-bit words, or null): one bit per 2^wshift filter slots -- "a slot of this group has been written by
  // this search".  A neighbour whose bit is clear finds an entry of an older epoch (or none) whatever the table holds: its probe --
  // a 4-byte read of a random 128-byte line of a table of up to 2 MiB -- is not issued.  Under load these searches are bound by
  // such lines, and nine probes in ten of a long search find an empty slot.
  prof = WANN_PROF_PTR(prof);
  // (every wave-uniform argument into scalar registers: see wave_beam_search_big)
  tag = (uint32_t)uni((int)tag);
  B = uni(B);
  bits = uni(bits);
  qid = uni64(qid);
  degree_limit = uni(degree_limit);
  mini_mask = (uint32_t)uni((int)mini_mask);
  wshift = uni(wshift);
  const int lane = lane_id();
  const uint32_t tmask = (1u << bits) - 1u;
  const int64_t row_off = uni(part.start);
  const int64_t row_base = uni64(part.row_base);
  const int rs = uni(ix.rs);
  u64 *const mb = L.lbeam;
  if (wbits)
    for (int i = lane; i < wwords; i += 64) wbits[i] = 0u;
  const int mode = NBC > 0 ? NBC : (RowRegsFor<METRIC>::NR == 0) ? 0 : uni(row_regs_blocks<METRIC>(ix));  // (blocks of a row a lane holds; 0: rows fetched where they are scored)
  const int lim = uni(limit > 0x7fffffff ? 0x7fffffff : (int)limit);

  // frontier = {start node 0} (beamSearch.h:80-82)
  if (lane == 0) L.cand_id[0] = 0;
  WAVE_SYNC();
  float d0 = wave_distances<METRIC>(ix, L.cand_id, L.cand_dist, L.qv, 1, row_off);
  d0 = __shfl(d0, 0);
  const u64 key0 = (u64)fkey(d0) << 32;
  if (lane == 0) mb[0] = key0;
  WAVE_SYNC();
  int M = 1, nvis = 0, ncmp_v = 0;  // entries of the LDS beam; hops; dist_cmps (counted per lane, summed at the end)
  // PENDING candidates: pend[0 .. D), pmin = the smallest of their keys (~0: none).  The beam is the LDS beam united with
  // them, truncated to B; the union is deferred (see above).  (pend: the 512 bytes of cand_id + cand_dist, which this core
  // uses for the start node only; cand_key is the union's scratch and may double as the slot-sharing test's)
  u64 *const pend = reinterpret_cast<u64 *>(L.cand_id);
  int D = 0;
  u64 pmin = ~0ull;
  float cutoff = 2147483648.0f;  // (float)INT_MAX while the LDS beam is not full, else its last distance: never below the true cutoff
  float lbv = 2147483648.0f;     // never above the true cutoff (set_lower_bound); equal to `cutoff` when nothing is pending
  // window: wv = mb[wbase + lane]; bit i of wum: entry wbase + i exists and is unvisited
  int wbase = 0;
  u64 wv = lane == 0 ? key0 : 1ull, wum = 1ull;
  // the hop in flight: node at position pos_c, its row `a`, the neighbours the reference scores (`kept`), their vectors requested;
  // hop_exact: the row lists a node twice within a filter-slot class (its candidates may hold a key twice: united at once, multiset rule)
  int pos_c = -1, a = -1, sc_r = 0, sc_nt = 0, sc_touch = 0, scan_from = 0;
  bool kept = false, have = false, hop_exact = false;
  typename RowRegsFor<METRIC>::type rr;
  // the next two expected nodes: s1 (key s1k at position s1p; row s1a; filter slots s1loc; probes s1old, valid while s1probe;
  // s1clash: two of its neighbours share a filter slot) and s2 (row s2a)
  int s1n = -1, s1p = 0, s2n = -1, s1a = -1, s2a = -1, s1old = 0;
  u64 s1k = 0;
  uint32_t s1loc = 0;
  bool s1probe = false, s1clash = false;

  auto load_row = [&](int node) -> int {  // graph.h:198; -1 = unused slot
    int v = -1;
    if (lane < rs) v = ix.graph[(row_base + node) * (int64_t)rs + lane];
    return v;
  };
  auto is_valid = [&](int arow) -> bool { return (arow >= 0) && (lane < degree_limit) && ((int64_t)arow != qid); };
  // what the filter holds in a neighbour's slot (-1: nothing this search has written -- never equal to a tagged id)
  auto probe = [&](int arow, uint32_t loc) -> int {
    bool maybe = is_valid(arow);
    if (wbits) {
      const uint32_t g = loc >> wshift;
      maybe = maybe && ((wbits[g >> 5] >> (g & 31u)) & 1u);
    }
    int old = -1;
    if (maybe) old = gtable[loc];
    return old;
  };
  auto mark_written = [&](bool stores, uint32_t loc) {
    if (wbits && stores) {
      const uint32_t g = loc >> wshift;
      __hip_atomic_fetch_or(wbits + (g >> 5), 1u << (g & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
  };
  // filter slots of a row + the exact test "two valid lanes of the row share a filter slot" (see wave_beam_search_big)
  auto prepare = [&](int arow, uint32_t &loc, bool &clash) {
    loc = (uint32_t)hash64_2((u64)(uint32_t)arow) & tmask;
    const bool valid = is_valid(arow);
    const uint32_t mh = loc & mini_mask;
    if (valid) mini[mh] = lane;
    WAVE_SYNC();
    const int mw = valid ? mini[mh] : lane;
    WAVE_SYNC();
    const uint32_t loc_w = (uint32_t)__shfl((int)loc, mw);
    const bool lost = valid && (mw != lane);
    bool c = ballot64(lost && loc_w == loc) != 0;
    for (u64 um = ballot64(lost && loc_w != loc); um && !c; um &= um - 1) {
      const int u = ctz64(um);
      const uint32_t lu = (uint32_t)rdlane((int)loc, u);
      c = ballot64(valid && loc == lu && lane != u) != 0;
    }
    clash = uni((int)c) != 0;
  };
  // lossy seen-filter with the reference's sequential semantics (beamSearch.h:68-73,126-131): what the reference scores.
  // twice: the row lists one node in two lanes of a slot class (the reference's builder can append the start point twice) -- only
  // then can the hop's kept candidates hold one key twice, and its union needs the multiset rule
  auto filter = [&](int arow, uint32_t loc, int old, bool clash, bool &twice) -> bool {
    const bool valid = is_valid(arow);
    const int tagged = (int)(tag | (uint32_t)arow);
    bool seen;
    twice = false;
    if (WANN_LIKELY(!clash)) {
      // (a slot that holds this id already is left alone: the store would change nothing, and its line -- one 128-byte line of
      // HBM per neighbour: the filter of a beam-1 280 search is 2 MiB -- need not be written back.  Under load these searches
      // are bound by such traffic, not by latency: half of what a hop moves is filter lines.)
      seen = valid && (old == tagged);
      if (valid && !seen) gtable[loc] = tagged;
      mark_written(valid && !seen, loc);
    } else {  // exact emulation: the nearest preceding lane of the same slot, else the table; the last lane of a slot class stores
      u64 eq = ballot64(valid);
      for (int b = 0; b < bits; b++) {
        const bool bit = (loc >> b) & 1u;
        const u64 bm = ballot64(valid && bit);
        eq &= bit ? bm : ~bm;
      }
      const u64 lower = valid ? (eq & lanemask_lt()) : 0ull;
      const u64 higher = (lane == 63) ? 0ull : (eq >> (lane + 1));
      const int prev_lane = lower ? (63 - __builtin_clzll(lower)) : lane;
      const int prev_val = __shfl(arow, prev_lane);
      seen = valid && (lower ? (prev_val == arow) : (old == tagged));
      if (valid && higher == 0) gtable[loc] = tagged;
      mark_written(valid && higher == 0, loc);
      // does any lane's id repeat an earlier lane's of its slot class?
      bool tw = false;
      u64 lm = lower;
      while (ballot64(lm != 0)) {
        const int l = lm ? (63 - __builtin_clzll(lm)) : lane;
        const int v = __shfl(arow, l);
        if (lm && v == arow) tw = true;
        if (lm) lm &= ~((u64)1 << l);
      }
      twice = uni((int)(ballot64(tw) != 0)) != 0;
    }
    const bool k = valid && !seen;
    ncmp_v += k ? 1 : 0;
    return k;
  };
  auto set_cutoff = [&]() {
    cutoff = 2147483648.0f;
    if (M >= B) cutoff = funkey((uint32_t)(mb[M - 1] >> 32));
  };
  // lbv (every lane the same value): a LOWER bound of the reference's cutoff while candidates are pending.  The reference's
  // B-th entry is an entry of (LDS beam U pending); at most D pending keys sort before it, so it lies at or behind position
  // B - 1 - D of the LDS beam.  No pending candidates, or fewer than B entries in all: the stale cutoff is the exact one.
  auto set_lower_bound = [&]() {
    lbv = cutoff;
    if (D > 0 && M + D >= B) lbv = (B - 1 - D >= 0) ? funkey((uint32_t)(mb[B - 1 - D] >> 32)) : -__builtin_inff();
  };
  // the pending candidates into the LDS beam (std::set_union + truncate, beamSearch.h:148-157, for all their hops at once)
  auto unite_pending = [&]() {
    if (D == 0) return;
    const u64 k = lane < D ? pend[lane] : 0ull;
    WAVE_SYNC();
    int p0;
    M = wave_merge<u64 *, true, false, true>(mb, M, B, lane < D, k, L.cand_key, &p0);
    scan_from = p0 < scan_from ? p0 : scan_from;
    D = 0;
    pmin = ~0ull;
    set_cutoff();
  };
  auto load_window = [&](int from) {
    wbase = from;
    const int x = from + lane;
    wv = x < M ? mb[x] : 1ull;
    wum = ballot64(!(wv & 1ull));
  };
  // The first two unvisited entries of the window become s1 / s2.  Rows on hand are kept (by node: a row is a pure function of
  // it), the others requested; s1's probes go out if its row is on hand (they follow the current hop's filter stores; a row
  // requested just now returns behind whatever is in flight: its probes go out when that is consumed).  Nothing is in flight
  // when this runs: registers move freely.
  auto expect = [&]() {
    if (WANN_UNLIKELY(popc64(wum) < 2 && wbase + 64 < M)) load_window(wum ? wbase + ctz64(wum) : wbase + 64);
    int g1 = -1, g2 = -1, p1 = 0;
    u64 k1 = 0;
    if (wum) {
      const int i = ctz64(wum);
      k1 = rdlane64(wv, i);
      g1 = (int)((uint32_t)k1 >> 1);
      p1 = wbase + i;
      const u64 rest = wum & (wum - 1);
      if (rest) g2 = (int)((uint32_t)rdlane((int)(uint32_t)wv, ctz64(rest)) >> 1);
    }
    // (Order matters to the compiler's wait insertion: a use of a register that MAY be the target of a request just issued -- on
    // any path -- waits for every request before it.  So: first everything that works on rows on hand, then the new requests.)
    const int o1n = s1n, o2n = s2n;
    const int o1a = s1a, o2a = s2a;
    const bool hand1 = g1 >= 0 && (g1 == o1n || g1 == o2n);
    if (g1 != o1n) {
      s1probe = false;
      if (g1 == o2n) s1a = o2a;
    }
    const bool hand2 = g2 >= 0 && (g2 == o2n || g2 == o1n);
    if (g2 == o1n && g2 != o2n) s2a = o1a;
    s1n = g1;
    s1k = k1;
    s1p = p1;
    s2n = g2;
    if (hand1 && !s1probe) {
      prepare(s1a, s1loc, s1clash);
      s1old = probe(s1a, s1loc);
      s1probe = true;
    }
    if (g1 >= 0 && !hand1) s1a = load_row(g1);
    if (g2 >= 0 && !hand2) s2a = load_row(g2);
  };

  unsigned long long tp = 0, acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define WANN_PHASE(i)                                       \
  do {                                                      \
    if (prof) {                                             \
      unsigned long long tn = __builtin_readcyclecounter(); \
      acc[i] += tn - tp;                                    \
      tp = tn;                                              \
    }                                                       \
  } while (0)
  if (prof) tp = __builtin_readcyclecounter();
  for (;;) {
    // (the loop-carried scalars, declared uniform once per hop: see wave_beam_search_big)
    M = uni(M);
    D = uni(D);
    nvis = uni(nvis);
    pos_c = uni(pos_c);
    scan_from = uni(scan_from);
    wbase = uni(wbase);
    s1n = uni(s1n);
    s1p = uni(s1p);
    s2n = uni(s2n);
    wum = (u64)uni64((long long)wum);
    pmin = (u64)uni64((long long)pmin);
    s1k = (u64)uni64((long long)s1k);
    s1probe = uni((int)s1probe) != 0;
    s1clash = uni((int)s1clash) != 0;
    have = uni((int)have) != 0;
    hop_exact = uni((int)hop_exact) != 0;
    cutoff = __builtin_bit_cast(float, uni(__builtin_bit_cast(int, cutoff)));
    lbv = __builtin_bit_cast(float, uni(__builtin_bit_cast(int, lbv)));

    bool moved = false;  // the LDS beam changed in this iteration: positions and the window are stale
    bool pass = false;   // the finished hop's candidates (key / pass / pm): appended to the pending ones further down
    u64 key = 0, pm = 0;
    if (WANN_LIKELY(have)) {
      // ---- the hop in flight: distances (beamSearch.h:135-145); what passes joins the pending candidates
      const float dist = mid_take_distances<METRIC, NBC>(ix, a, kept, row_off, L.qv, mode, rr, sc_r, sc_nt, sc_touch);
      pass = kept && (dist < cutoff);
      // `cutoff` is the LDS beam's: stale while candidates are pending.  A candidate in [true cutoff, cutoff) is admitted though the
      // reference rejects it (beamSearch.h:135-145: dist >= cutoff); it does no harm when it sorts BEHIND the reference's B-th
      // entry -- never visited, gone at the next truncation -- i.e. unless its distance EQUALS that entry's and its id is
      // smaller (tie-heavy data: small integer coordinates, duplicate rows).  The B-th entry is one of the LDS beam's entries
      // from position B - 1 - D on or a pending one: a candidate at or above the lower bound whose distance equals one of
      // theirs sends the hop down the exact path -- the pending candidates are united first and the test is repeated against
      // the exact cutoff.
      const u64 amb = ballot64(pass && dist >= lbv);
      if (WANN_UNLIKELY(amb != 0)) {
        WAVE_SYNC();
        const float qn = __builtin_nanf("");
        const int from = B - 1 - D > 0 ? B - 1 - D : 0;
        const float pd = lane < D ? funkey((uint32_t)(pend[lane] >> 32)) : qn;
        const float t0 = from + lane < M ? funkey((uint32_t)(mb[from + lane] >> 32)) : qn;
        const float t1 = from + 64 + lane < M ? funkey((uint32_t)(mb[from + 64 + lane] >> 32)) : qn;
        bool tie = false;
        for (u64 mm = amb; mm; mm &= mm - 1) {
          const float dc = __builtin_bit_cast(float, rdlane(__builtin_bit_cast(int, dist), ctz64(mm)));
          tie = tie || pd == dc || t0 == dc || t1 == dc;
        }
        if (ballot64(tie) != 0) {
          unite_pending();
          moved = true;
          pass = kept && (dist < cutoff);
        }
      }
      key = ((u64)fkey(dist) << 32) | ((u64)(uint32_t)a << 1);
      pm = ballot64(pass);
      WANN_PHASE(0);  // vectors + distances
      if (WANN_UNLIKELY(hop_exact)) {
        // (this row lists a node twice within a slot class -- the multiset union may keep two copies of its key: the hop's
        // candidates are united on their own)
        unite_pending();
        int p0;
        M = wave_merge(mb, M, B, pass, key, L.cand_key, &p0);
        scan_from = p0 < scan_from ? p0 : scan_from;
        set_cutoff();
        moved = true;
        pm = 0;
      } else if (pm) {
        if (WANN_UNLIKELY(D + popc64(pm) > 64)) {
          unite_pending();
          moved = true;
        }
        // (they count as pending from here on; their keys reach the buffer behind the next hop's requests)
        for (u64 mm = pm; mm; mm &= mm - 1) {
          const u64 kc = rdlane64(key, ctz64(mm)) | 1ull;
          pmin = kc < pmin ? kc : pmin;
        }
      }
      WANN_PHASE(2);  // pending / union
      // (a row that was requested behind these vectors is back with them: its probes can go out now)
      if (s1n >= 0 && !s1probe) {
        prepare(s1a, s1loc, s1clash);
        s1old = probe(s1a, s1loc);
        s1probe = true;
      }
    }
    auto append = [&]() {  // the finished hop's candidates into the pending buffer
      if (pm) {
        if (pass) pend[D + popc64(pm & lanemask_lt())] = key;
        D += popc64(pm);
      }
    };
    // ---- The next hop visits s1 -- the first unvisited entry of the LDS beam -- unless a pending candidate sorts at or before
    //      it (an equal key is a copy of s1's entry).  Then s1 is the closest unvisited entry of the whole beam and lies within
    //      its first B entries: s1's rank is its position.
    const bool committed = have && !moved && s1n >= 0 && nvis < lim && pmin > (s1k | 1ull);
    int a_nx;
    bool kept_nx;
    if (WANN_LIKELY(committed)) {
      if (lane == 0) mb[s1p] = s1k | 1ull;
      wum &= ~((u64)1 << (s1p - wbase));
      nvis++;
      pos_c = s1p;
      scan_from = s1p + 1;
      a_nx = s1a;
      kept_nx = filter(s1a, s1loc, s1old, s1clash, hop_exact);
      if (prof) acc[5]++;
    } else {
      // ---- exact beam first; then the closest unvisited entry (beamSearch.h:108-117): entries before scan_from are visited
      append();
      unite_pending();
      WANN_PHASE(2);
      if (nvis >= lim) break;
      int cur = -1;
      u64 ck = 0;
      for (int sp = scan_from; sp < M; sp += 64) {
        const int x = sp + lane;
        const u64 e = x < M ? mb[x] : 1ull;
        const u64 um = ballot64(!(e & 1ull));
        if (um) {
          const int i = ctz64(um);
          ck = rdlane64(e, i);
          cur = (int)((uint32_t)ck >> 1);
          pos_c = sp + i;
          break;
        }
      }
      cur = uni(cur);
      pos_c = uni(pos_c);
      if (cur < 0) break;
      if (lane == 0) mb[pos_c] = ck | 1ull;
      nvis++;
      scan_from = pos_c + 1;
      // row and probes: on hand if this is an expected node (probes: only s1's, and only if no other hop's stores followed them)
      uint32_t loc;
      int old;
      bool clash;
      if (cur == s1n) {
        a_nx = s1a;
        if (s1probe) {
          loc = s1loc;
          old = s1old;
          clash = s1clash;
        } else {
          prepare(a_nx, loc, clash);
          old = probe(a_nx, loc);
        }
      } else {
        a_nx = (cur == s2n) ? s2a : load_row(cur);
        prepare(a_nx, loc, clash);
        old = probe(a_nx, loc);
      }
      kept_nx = filter(a_nx, loc, old, clash, hop_exact);
      load_window(pos_c + 1);
      WANN_PHASE(3);  // unexpected node / stale positions: union, scan, row, probes, filter
    }
    s1probe = false;  // (this hop's stores follow whatever probes are out)
    WANN_PHASE(1);    // filter
    // ---- this hop's vectors FIRST (what the next iteration waits for), then the expectations -- s1's probes (they follow this
    //      hop's filter stores), the rows of s1 / s2 -- and, behind all requests, the finished hop's candidates into the pending
    //      buffer: whatever is done between a hop's requests and the next hop's costs nothing while the vectors travel
    sc_touch = 0;
    mid_request_rows<METRIC, NBC>(ix, a_nx, kept_nx, row_off, mode, rr, sc_r, sc_nt, sc_touch);
    WANN_PHASE(6);  // vector requests
    expect();
    if (committed) append();
    set_lower_bound();
    a = a_nx;
    kept = kept_nx;
    have = true;
    WANN_PHASE(4);  // expectations, probes, pending buffer
  }
#undef WANN_PHASE
  for (int o = 32; o; o >>= 1) ncmp_v += __shfl_xor(ncmp_v, o);
  if (prof && lane == 0)
    for (int i = 0; i < 7; i++) atomicAdd(&prof[i], acc[i]);  // (5: hops committed without a union)
  m_out = M;
  nvis_out = nvis;
  ncmp_out = 1 + ncmp_v;
}

// --------------------------------------------------------------------------------------------
// Register-resident variant (B <= 64 * NE, seen-filter in LDS): the beam lives in registers for the
// whole search, entry x in lane x % 64, slot x / 64.  The union with the scored candidates is
// computed by ONE loop over the passing candidates in which every lane compares its beam entries
// and its own candidate with the broadcast candidate key (ballots give the insertion point and
// the duplicate test as scalars), so there is no binary search and a single LDS round trip per
// hop (write the merged beam, read it back).  Same results as wave_beam_search (the parity tests
// run both).
// --------------------------------------------------------------------------------------------
template <int METRIC, int NE>
__device__ __forceinline__ void wave_beam_search_small(const IndexView &ix, const PartDesc &part, const WaveLds &L,
                                                       int B, int bits, int64_t qid, int64_t limit, int degree_limit,
                                                       int &m_out, long long &nvis_out, long long &ncmp_out,
                                                       unsigned long long *prof = nullptr) {
  prof = WANN_PROF_PTR(prof);
  const int lane = lane_id();
  const uint32_t tmask = (1u << bits) - 1u;
  constexpr uint32_t TAG = 0x80000000u;
  const int64_t row_off = part.start;
  {  // (16-byte aligned: carve_wave_lds)
    int4 *lt = reinterpret_cast<int4 *>(L.ltable);
    for (int i = lane; i < (1 << (bits - 2)); i += 64) lt[i] = make_int4(-1, -1, -1, -1);
  }
  if (lane == 0) L.cand_id[0] = 0;
  WAVE_SYNC();
  float d0 = wave_distances<METRIC>(ix, L.cand_id, L.cand_dist, L.qv, 1, row_off);
  d0 = __shfl(d0, 0);
  int m = 1, p = 0;
  long long nvis = 0, ncmp = 1;
  u64 e[NE];  // beam entries of this lane: e[j] is entry 64*j + lane (valid while < m), else ~0
#pragma unroll
  for (int j = 0; j < NE; j++) e[j] = ~0ull;
  if (lane == 0) e[0] = (u64)fkey(d0) << 32;
  unsigned long long tp = 0, acc[6] = {0, 0, 0, 0, 0, 0}, nc_total = 0;
#define WANN_PHASE(i)                                       \
  do {                                                      \
    if (prof) {                                             \
      unsigned long long tn = __builtin_readcyclecounter(); \
      acc[i] += tn - tp;                                    \
      tp = tn;                                              \
    }                                                       \
  } while (0)
  if (prof) tp = __builtin_readcyclecounter();
  auto entry = [&](int x) -> u64 {  // wave-uniform x
    u64 v = 0;
#pragma unroll
    for (int j = 0; j < NE; j++)
      if ((x >> 6) == j) v = rdlane64(e[j], x & 63);
    return v;
  };

  while (p < m && nvis < limit) {
    const u64 curkey = entry(p);
    const int cur = (int)((uint32_t)curkey >> 1);
#pragma unroll
    for (int j = 0; j < NE; j++)
      if (64 * j + lane == p) e[j] |= 1ull;
    nvis++;
    int a = -1;
    if (lane < ix.rs) a = ix.graph[(part.row_base + cur) * (int64_t)ix.rs + lane];
    const bool valid = (a >= 0) && (lane < degree_limit) && ((int64_t)a != qid);
    if (prof) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    WANN_PHASE(0);

    // ---- seen-filter.  Fast path: tag every slot with the lane number; if every lane reads its own tag
    //      back, no two lanes of the row share a slot and the sequential rule is just "old == id".
    const uint32_t loc = (uint32_t)hash64_2((u64)(uint32_t)a) & tmask;
    int old = -1;
    if (valid) old = L.ltable[loc];
    WAVE_SYNC();
    if (valid) L.ltable[loc] = (int)(TAG | (uint32_t)lane);
    WAVE_SYNC();
    int rb = 0;
    if (valid) rb = L.ltable[loc];
    const bool clash = valid && (rb != (int)(TAG | (uint32_t)lane));
    bool seen;
    const bool had_clash = ballot64(clash) != 0;
    if (!had_clash) {
      seen = valid && (old == a);
      WAVE_SYNC();
      if (valid) L.ltable[loc] = a;
    } else {  // shared slots (or a node listed twice): exact sequential emulation
      u64 eq = ballot64(valid);
      for (int b = 0; b < bits; b++) {
        const bool bit = (loc >> b) & 1u;
        const u64 bm = ballot64(valid && bit);
        eq &= bit ? bm : ~bm;
      }
      const u64 lower = eq & lanemask_lt();
      const u64 higher = (lane == 63) ? 0ull : (eq >> (lane + 1));
      const int prev_lane = lower ? (63 - __builtin_clzll(lower)) : lane;
      int prev_val = __shfl(a, prev_lane);
      if (!lower) prev_val = old;
      seen = valid && (prev_val == a);
      WAVE_SYNC();
      if (valid) L.ltable[loc] = a;               // clears every tag; classes end up holding ...
      WAVE_SYNC();
      if (valid && higher == 0) L.ltable[loc] = a;  // ... the id of their last lane
    }
    const bool keep = valid && !seen;
    const u64 kmask = ballot64(keep);
    const int nk = popc64(kmask);
    if (keep) L.cand_id[popc64(kmask & lanemask_lt())] = a;
    WAVE_SYNC();
    ncmp += nk;
    WANN_PHASE(1);

    float cutoff = 2147483648.0f;
    if (m >= B) cutoff = funkey((uint32_t)(entry(m - 1) >> 32));
    const float dist = wave_distances<METRIC>(ix, L.cand_id, L.cand_dist, L.qv, nk, row_off);
    const int cid = (lane < nk) ? L.cand_id[lane] : 0;
    WAVE_SYNC();
    const bool pass = (lane < nk) && (dist < cutoff);
    const u64 key = ((u64)fkey(dist) << 32) | ((u64)(uint32_t)cid << 1);
    WANN_PHASE(2);

    // ---- union (std::set_union multiset rule) with every operand in registers
    const u64 smask = ballot64(pass);
    if (smask) {
      u64 ek[NE];  // empty slots hold ~0: never below a candidate
      int sx[NE];
#pragma unroll
      for (int j = 0; j < NE; j++) {
        ek[j] = e[j] | 1ull;
        sx[j] = 0;
      }
      const u64 kk = key | 1ull;
      int rank = 0, mypos = 0, cp = 0;
      bool mydup = false;
      for (u64 mm = smask; mm; mm &= mm - 1) {
        const int i = ctz64(mm);
        const u64 ki = rdlane64(kk, i);
        bool below[NE];
        int pos_i = 0;
        u64 eqb = 0;
        int neq = 0;
#pragma unroll
        for (int j = 0; j < NE; j++) {
          below[j] = ek[j] < ki;  // my beam entry sorts before candidate i
          pos_i += popc64(ballot64(below[j]));
          const u64 eqj = ballot64(ek[j] == ki);
          eqb |= eqj;
          neq += popc64(eqj);
        }
        bool dup_i, before_me;
        if (!had_clash) {
          // the row held distinct ids in distinct filter slots, so candidate keys are distinct and
          // candidate i is a copy only if the beam already holds its key
          dup_i = eqb != 0;
          before_me = pass && (ki < kk);
        } else {  // general multiset rule of std::set_union
          const int ji = popc64(ballot64(pass && kk == ki) & (((u64)1 << i) - 1));
          dup_i = ji < neq;
          before_me = pass && (ki < kk || (ki == kk && i < lane));
        }
        if (lane == i) {
          mypos = pos_i;
          mydup = dup_i;
        }
        if (!dup_i) {
          cp++;
          rank += before_me ? 1 : 0;
#pragma unroll
          for (int j = 0; j < NE; j++) sx[j] += (!below[j]) ? 1 : 0;  // (empty slots are never written back)
        }
      }
      if (prof) {
        unsigned long long tn = __builtin_readcyclecounter();
        acc[5] += tn - tp;  // merge loop only (not reset: phase 3 still covers everything)
        nc_total += popc64(smask);
      }
      if (cp) {
#pragma unroll
        for (int j = 0; j < NE; j++) {
          const int x = 64 * j + lane;
          if (x < m) {
            const int nx = x + sx[j];
            if (nx < B) L.lbeam[nx] = e[j];
          }
        }
        if (pass && !mydup) {
          const int np = mypos + rank;
          if (np < B) L.lbeam[np] = key;
        }
        WAVE_SYNC();
        m = (m + cp) < B ? (m + cp) : B;
#pragma unroll
        for (int j = 0; j < NE; j++) e[j] = (64 * j + lane < m) ? L.lbeam[64 * j + lane] : ~0ull;
        WAVE_SYNC();
      }
    }
    WANN_PHASE(3);
    p = m;
#pragma unroll
    for (int j = NE - 1; j >= 0; j--) {
      const u64 um = ballot64((64 * j + lane < m) && !(e[j] & 1ull));
      if (um) p = 64 * j + ctz64(um);
    }
    WANN_PHASE(4);
  }
#undef WANN_PHASE
#pragma unroll
  for (int j = 0; j < NE; j++)
    if (64 * j + lane < m) L.lbeam[64 * j + lane] = e[j];  // final beam for the caller
  WAVE_SYNC();
  if (prof && lane == 0) {
    for (int i = 0; i < 6; i++) atomicAdd(&prof[i], acc[i]);
    atomicAdd(&prof[6], nc_total);
  }
  m_out = m;
  nvis_out = nvis;
  ncmp_out = ncmp;
}

}  // namespace wann
