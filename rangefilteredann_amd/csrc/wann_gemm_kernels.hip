// wann_gemm_kernels.hip -- the dense prefilter path: when many queries share one label window
// (PrefilterIndex::batch_search, src/prefiltering.h:124-204 -- e.g. the adversarial dataset, where 99
// queries share each 10 000-point window) the brute-force scan is a true Q x N contraction and runs
// on the matrix cores.  Everything is planned and run on the device; the host only enqueues:
//
//   k_group_clear / k_group_insert / k_group_plan / k_group_scatter
//                    group the batch's queries by window (open-addressing table over (a, b)), lay out the groups'
//                    query lists and the tile list, hand the ungrouped queries to the exact scan
//   k_gemm_select    per (window group, 128-query tile, window slice): S = Q . P^T on v_mfma_f32_32x32x16_bf16
//                    with both operands split into two bf16 terms (q = q1 + q2 + ..., three products: q1 p1 + q1 p2 +
//                    q2 p1, fp32 accumulate), scores -q.p (MIPS) or |p|^2 - 2 q.p (L2, |q|^2 added later), and --
//                    fused into the epilogue -- the selection of the best ~32 scores per query: a score below the
//                    query's running threshold is appended to a 64-slot LDS list; full lists are cut back to the
//                    best 32..38 by a ballot quick-select, which also lowers the threshold
//   k_rerank         per query: the 32 best candidates over its slices' lists, their exact reference-order distances,
//                    ordered by (dist, id), first k; plus a proof that no unselected point can belong to the top k
//                    (score error bound); queries that cannot be proven go to the exact scan kernel k_brute
//
// The MFMA scores only SELECT candidates (the reference sums in another order); every returned
// distance is computed by the reference-order routines.  No score matrix is ever written to memory.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "wann_gemm_device.h"
#include "wann_wave.h"

namespace wann {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2g __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned long long kEmptySlot = ~0ull;
constexpr float kInf = __builtin_inff();

__global__ void k_point_norms(IndexView ix, float *norm2, unsigned int *max_bits) {
  const int lane = lane_id();
  const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= ix.n) return;
  const float *p = ix.points + row * (int64_t)ix.stride;
  float s = 0.f;
  for (int i = lane; i < ix.d; i += 64) s = fmaf(p[i], p[i], s);
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) {
    norm2[row] = s;
    // s >= 0: the bit pattern orders like the value; only a new maximum pays for the atomic
    if (__float_as_uint(s) > __hip_atomic_load(max_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(max_bits, __float_as_uint(s));
  }
}

// ------------------------------------------------------------------------------------------------
// grouping
// ------------------------------------------------------------------------------------------------
__global__ void k_group_clear(GemmArgs A) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i <= A.cap_mask) {
    A.slot_key[i] = kEmptySlot;
    A.slot_count[i] = 0;
  }
  if (i < P_INTS) A.plan[i] = 0;
}

__global__ void k_group_insert(GemmArgs A) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= A.nq) return;
  const Task t = A.tasks[q];
  if (t.mode != T_BRUTE_GATHER) {
    A.q_slot[q] = -1;
    return;
  }
  const unsigned long long key = ((unsigned long long)(uint32_t)t.a << 32) | (uint32_t)t.b;
  uint32_t pos = (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> 40) & (uint32_t)A.cap_mask;
  for (;;) {
    const unsigned long long old = atomicCAS(&A.slot_key[pos], kEmptySlot, key);
    if (old == kEmptySlot || old == key) break;
    pos = (pos + 1) & (uint32_t)A.cap_mask;
  }
  A.q_slot[q] = (int32_t)pos;
  A.q_rank[q] = atomicAdd(&A.slot_count[pos], 1);
}

// one workgroup: every occupied slot becomes a group (or is left to the exact scan), with its share of the query
// list and of the tile list
__global__ __launch_bounds__(1024) void k_group_plan(GemmArgs A, Counters *ctr) {
  if (threadIdx.x == 0) *A.brute_count = 0;  // k_group_scatter rebuilds the exact-scan list
  for (int pos = threadIdx.x; pos <= A.cap_mask; pos += blockDim.x) {
    const unsigned long long key = A.slot_key[pos];
    if (key == kEmptySlot) continue;
    const int qc = A.slot_count[pos];
    const int64_t a = (int64_t)(key >> 32), b = (int64_t)(key & 0xffffffffull), w = b - a;
    if (qc < kGroupMinQueries || w < kGroupMinWindow) {
      A.slot_group[pos] = -1;
      continue;
    }
    const int g = atomicAdd(&A.plan[P_NGROUPS], 1);
    GemmGroup G;
    G.a = a;
    G.b = b;
    G.qoff = atomicAdd(&A.plan[P_NTQ], qc);
    G.qcount = qc;
    int64_t cs = kGemmPointChunk;
    if ((w + cs - 1) / cs > kMaxChunks) cs = (((w + kMaxChunks - 1) / kMaxChunks) + 127) & ~(int64_t)127;
    G.chunk = (int32_t)cs;
    G.nch = (int32_t)((w + cs - 1) / cs);
    const int nqt = (qc + 127) >> 7;
    const int t0 = atomicAdd(&A.plan[P_NTILES], nqt * G.nch);
    for (int ch = 0; ch < G.nch; ch++)
      for (int qt = 0; qt < nqt; qt++) A.tiles[t0 + ch * nqt + qt] = GemmTile{g, qt << 7, ch};
    A.groups[g] = G;
    A.slot_group[pos] = g;
  }
  __syncthreads();
  if (threadIdx.x == 0) ctr->gemm_queries = (unsigned long long)A.plan[P_NTQ];
}

__global__ void k_group_scatter(GemmArgs A) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= A.nq) return;
  const int pos = A.q_slot[q];
  if (pos < 0) return;
  const int g = A.slot_group[pos];
  if (g < 0) {
    A.brute_list[atomicAdd(A.brute_count, 1)] = (int32_t)q;  // stand-alone PrefilterIndex: one task slot per query
    return;
  }
  const int tq = A.groups[g].qoff + A.q_rank[q];
  A.gq[tq] = (int32_t)q;
  A.tq_group[tq] = g;
  A.tq_local[tq] = A.q_rank[q];
  A.thr_shared[tq] = 0xffffffffu;
}

// ------------------------------------------------------------------------------------------------
// GEMM + selection
// ------------------------------------------------------------------------------------------------
// two floats -> two bf16 (round to nearest even, v_cvt_pk_bf16_f32) and the bf16 of what the rounding left:
// a = hi + lo + r with |r| <= 2^-16 |a| (|a - hi| <= 2^-8 |a| is exactly representable, so lo rounds it to 2^-8 again)
__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {
  const f32x2g v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ void split2(float a, float b, uint32_t &hi, uint32_t &lo) {
  hi = pk_bf16(a, b);
  lo = pk_bf16(a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u));
}

constexpr int kCutLow = 32, kCutHigh = 38;  // a full list is cut back to this many entries ...
constexpr int kCutTrigger = 56;             // ... once it holds this many

// The wave cuts row `r`'s candidate list (c <= 64 entries, unordered) back to the best kCutLow..kCutHigh: ballot
// quick-select for a pivot key with that many smaller keys; the pivot's score becomes the row's threshold (every
// dropped entry scores >= it).
__device__ __forceinline__ void cut_row(u64 *buf, float *thr, int *cnt, int r, int c, int lane) {
  const u64 key = (lane < c) ? buf[r * kCandCap + lane] : ~0ull;
  u64 act = ballot64(lane < c);
  int below = 0;  // keys known to lie below every active key
  u64 pk = 0, keep = 0;
  for (;;) {
    const int p = ctz64(act);
    pk = rdlane64(key, p);
    keep = ballot64(key < pk);
    const int cl = popc64(keep);
    if (cl >= kCutLow && cl <= kCutHigh) break;
    if (cl < kCutLow) {
      act &= ~keep & ~((u64)1 << p);
      below = cl + 1;
    } else
      act &= keep;
  }
  (void)below;
  const bool mine = (keep >> lane) & 1;
  const int dst = popc64(keep & lanemask_lt());
  if (mine) buf[r * kCandCap + dst] = key;
  if (lane == 0) {
    cnt[r] = popc64(keep);
    thr[r] = funkey((uint32_t)(pk >> 32));
  }
}

// One workgroup (4 waves, one per SIMD) per tile = (group, 128 queries, one slice of the window); tiles are taken
// round-robin by a grid of one workgroup per CU.  The wave owns a 64 x 64 corner of each 128 x 128 score block = 2 x 2
// MFMA tiles; its 64 query rows live in registers as bf16 pairs for the whole tile (A operand: 16 B per lane, k-step and
// term), the 128 staged points in the LDS as [hi | lo] bf16 rows (B operand: one ds_read_b128 feeds three MFMAs).
template <int STRIDE>  // padded row length in floats: a multiple of 16, <= 128
__global__ __launch_bounds__(256) void k_gemm_select(GemmArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const IndexView &ix = A.ix;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  constexpr int S = STRIDE / 16;       // MFMA k-steps per product term
  constexpr int RB = 4 * STRIDE + 16;  // bytes per staged point: hi row, lo row, 16 B so that 8 rows cover all banks
  unsigned char *Ps = smem;                                             // [128][RB]
  u64 *buf = reinterpret_cast<u64 *>(smem + 128 * RB);                  // [128][kCandCap] candidate keys
  float *thr = reinterpret_cast<float *>(buf + 128 * kCandCap);         // [128] per query row
  int *cnt = reinterpret_cast<int *>(thr + 128);                        // [128]
  float *base = reinterpret_cast<float *>(cnt + 128);                   // [128] per staged point: |p|^2 / 0 / +inf (beyond the slice)
  int *rid = reinterpret_cast<int *>(base + 128);                       // [128] point rows of the block being fetched
  constexpr int s4 = STRIDE >> 2;
  constexpr int nit = s4 >> 1;  // 128 rows x s4 float4 / 256 threads (s4 is even)
  const int half = lane >> 5, col = lane & 31, wr = wv >> 1, wc = wv & 1;
  const float scale = (ix.metric == 1) ? -1.f : -2.f;
  const int ntiles = A.plan[P_NTILES];

  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const GemmTile tile = A.tiles[t];
    const GemmGroup grp = A.groups[tile.group];
    const int64_t w = grp.b - grp.a, wlast = w - 1;
    const int q0 = tile.q0;
    const int64_t p_begin = (int64_t)tile.ch * grp.chunk;
    const int64_t p_end = (p_begin + grp.chunk < w) ? (p_begin + grp.chunk) : w;
    __syncthreads();  // the previous tile's lists have been written out
    if (tid < 128) {
      const bool live = q0 + tid < grp.qcount;
      thr[tid] = live ? kInf : -kInf;  // rows beyond the group take no candidates
      cnt[tid] = 0;
      rid[tid] = ix.fi_sorted[grp.a + min(p_begin + tid, wlast)];
    }
    // A operand: rows 64 wr + 32 i + col, columns 16 s + 8 half + (0..7).  The query tile passes through the LDS (where
    // the points will be staged) so that the global loads are coalesced; loads are unconditional (clamped indices,
    // select afterwards).
    u32x4 ah[2][S], al[2][S];
    {
      constexpr int DP = STRIDE + 4;  // 128 x DP floats = the staging area exactly
      float *Qs = reinterpret_cast<float *>(Ps);
      const int dlast = ix.d - 1;
#pragma unroll 2
      for (int it = 0; it < nit; it++) {
        const int idx = tid + it * 256;
        const int r = idx / s4, c = (idx - r * s4) * 4;
        const bool live = q0 + r < grp.qcount;
        const float *src = A.queries + (int64_t)A.gq[grp.qoff + (live ? q0 + r : grp.qcount - 1)] * ix.d;
        f32x4 v;
        v[0] = src[min(c + 0, dlast)]; v[1] = src[min(c + 1, dlast)]; v[2] = src[min(c + 2, dlast)]; v[3] = src[min(c + 3, dlast)];
#pragma unroll
        for (int e = 0; e < 4; e++) v[e] = (live && c + e < ix.d) ? v[e] : 0.f;
        *reinterpret_cast<f32x4 *>(Qs + r * DP + c) = v;
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int s = 0; s < S; s++) {
          const float *qp = Qs + (64 * wr + 32 * i + col) * DP + 16 * s + 8 * half;
          const f32x4 v0 = *reinterpret_cast<const f32x4 *>(qp), v1 = *reinterpret_cast<const f32x4 *>(qp + 4);
          uint32_t h, l;
          split2(v0[0], v0[1], h, l); ah[i][s][0] = h; al[i][s][0] = l;
          split2(v0[2], v0[3], h, l); ah[i][s][1] = h; al[i][s][1] = l;
          split2(v1[0], v1[1], h, l); ah[i][s][2] = h; al[i][s][2] = l;
          split2(v1[2], v1[3], h, l); ah[i][s][3] = h; al[i][s][3] = l;
        }
    }
    __syncthreads();
    // The next point block travels HBM -> registers while the MFMA loop of the current one runs (one wave per SIMD:
    // the 512-register budget is all ours), and registers -> bf16 pairs -> LDS after the barrier.  Its row numbers
    // were put in the LDS one step earlier, so no load depends on another load.
    f32x4 pre[nit];
    float pre_n = 0.f;
    int pre_rid = 0;
#define WANN_FETCH(C0)                                                                                     \
  {                                                                                                        \
    _Pragma("unroll") for (int it = 0; it < nit; it++) {                                                   \
      const int idx = tid + it * 256;                                                                      \
      const int r = idx / s4, c = (idx - r * s4) * 4;                                                      \
      pre[it] = *reinterpret_cast<const f32x4 *>(ix.points + (int64_t)rid[r] * STRIDE + c);                \
    }                                                                                                      \
    if (tid < 128) {                                                                                       \
      pre_n = A.pnorm2[rid[tid]];                                                                          \
      pre_rid = ix.fi_sorted[grp.a + min((C0) + 128 + tid, wlast)];                                        \
    }                                                                                                      \
  }
    WANN_FETCH(p_begin)
    f32x16 acc[2][2];
    u64 pend = 0;  // bit i*32 + j*16 + reg: a candidate that found its row's list full

    // the epilogue: scores below the row's threshold join the row's list
#define WANN_SCAN(RETRY, C0)                                                                               \
  {                                                                                                        \
    u64 still = 0;                                                                                         \
    const float bj2[2] = {base[64 * wc + col], base[64 * wc + 32 + col]};                                  \
    _Pragma("unroll") for (int i = 0; i < 2; i++) _Pragma("unroll") for (int g = 0; g < 4; g++) {          \
      const int rowb = 64 * wr + 32 * i + 8 * g + 4 * half;                                                \
      const f32x4 t4 = *reinterpret_cast<const f32x4 *>(thr + rowb);                                       \
      _Pragma("unroll") for (int j = 0; j < 2; j++) {                                                      \
        const int pc = 64 * wc + 32 * j + col;                                                             \
        const float bj = bj2[j];                                                                           \
        float sc[4];                                                                                       \
        bool ps[4];                                                                                        \
        _Pragma("unroll") for (int r = 0; r < 4; r++) {                                                    \
          sc[r] = fmaf(scale, acc[i][j][4 * g + r], bj);                                                   \
          ps[r] = sc[r] < t4[r];                                                                           \
          if (RETRY) ps[r] = ps[r] && ((pend >> (i * 32 + j * 16 + 4 * g + r)) & 1);                       \
        }                                                                                                  \
        if (ballot64(ps[0] | ps[1] | ps[2] | ps[3])) { /* rare: one test per four registers */             \
          _Pragma("unroll") for (int r = 0; r < 4; r++) if (ps[r]) {                                       \
            const int slot = atomicAdd(&cnt[rowb + r], 1);                                                 \
            if (slot < kCandCap)                                                                           \
              buf[(rowb + r) * kCandCap + slot] = ((u64)fkey(sc[r]) << 32) | (uint32_t)((C0) + pc);        \
            else                                                                                           \
              still |= (u64)1 << (i * 32 + j * 16 + 4 * g + r);                                            \
          }                                                                                                \
        }                                                                                                  \
      }                                                                                                    \
    }                                                                                                      \
    pend = still;                                                                                          \
  }
    // rows 32 wv .. 32 wv + 31 are this wave's to cut back
#define WANN_CUT_ROWS()                                                                                    \
  {                                                                                                        \
    const int myc = (lane < 32) ? cnt[32 * wv + lane] : 0;                                                 \
    u64 need = ballot64(myc >= kCutTrigger);                                                               \
    while (need) {                                                                                         \
      const int r = ctz64(need);                                                                           \
      need &= need - 1;                                                                                    \
      const int c = rdlane(myc, r);                                                                        \
      cut_row(buf, thr, cnt, 32 * wv + r, c < kCandCap ? c : kCandCap, lane);                              \
    }                                                                                                      \
  }

    for (int64_t c0 = p_begin; c0 < p_end; c0 += 128) {
      // (the barrier that ended the previous step's scan: nobody reads Ps / base / rid any more)
#pragma unroll
      for (int it = 0; it < nit; it++) {
        const int idx = tid + it * 256;
        const int r = idx / s4, c = (idx - r * s4) * 4;
        uint32_t h0, l0, h1, l1;
        split2(pre[it][0], pre[it][1], h0, l0);
        split2(pre[it][2], pre[it][3], h1, l1);
        *reinterpret_cast<uint2 *>(Ps + r * RB + c * 2) = make_uint2(h0, h1);
        *reinterpret_cast<uint2 *>(Ps + r * RB + 2 * STRIDE + c * 2) = make_uint2(l0, l1);
      }
      if (tid < 128) {
        base[tid] = (c0 + tid < p_end) ? ((ix.metric == 1) ? 0.f : pre_n) : kInf;
        rid[tid] = pre_rid;
      }
      __syncthreads();
      WANN_FETCH(c0 + 128)  // unconditional (row numbers are clamped): a conditional fetch would make the compiler wait for it here
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
          for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
      const unsigned char *pb = Ps + (64 * wc + col) * RB + 16 * half;
#pragma unroll
      for (int s = 0; s < S; s++) {
        bf16x8 bh[2], bl[2];
#pragma unroll
        for (int j = 0; j < 2; j++) {
          bh[j] = *reinterpret_cast<const bf16x8 *>(pb + j * 32 * RB + 32 * s);
          bl[j] = *reinterpret_cast<const bf16x8 *>(pb + j * 32 * RB + 2 * STRIDE + 32 * s);
        }
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
          for (int j = 0; j < 2; j++) {
            const bf16x8 a_hi = __builtin_bit_cast(bf16x8, ah[i][s]), a_lo = __builtin_bit_cast(bf16x8, al[i][s]);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, bl[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo, bh[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, bh[j], acc[i][j], 0, 0, 0);
          }
      }
      WANN_SCAN(false, c0)
      int any = __syncthreads_or(pend != 0);  // every wave is done with Ps; the appended keys are visible
      WANN_CUT_ROWS()
      while (any) {  // some list overflowed: retry those candidates against the lowered thresholds
        __syncthreads();
        WANN_SCAN(true, c0)
        any = __syncthreads_or(pend != 0);
        WANN_CUT_ROWS()
      }
    }
    __syncthreads();
    // hand the lists over
    for (int r = 32 * wv; r < 32 * wv + 32; r++) {
      if (q0 + r >= grp.qcount) break;
      const int64_t slot = (int64_t)(grp.qoff + q0 + r) * kMaxChunks + tile.ch;
      const int c = cnt[r];
      if (lane < c) A.cand_key[slot * kCandCap + lane] = buf[r * kCandCap + lane];
      if (lane == 0) {
        A.cand_cnt[slot] = c;
        A.cand_cut[slot] = thr[r];
      }
    }
  }
#undef WANN_FETCH
#undef WANN_SCAN
#undef WANN_CUT_ROWS
}

// one wave per grouped query: the kSelect best candidates of its slices' lists, exact distances, (dist, id) order, proof
template <int METRIC>
__global__ __launch_bounds__(256) void k_rerank(GemmArgs A, Counters *ctr) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const IndexView &ix = A.ix;
  const int lane = lane_id(), wv = threadIdx.x >> 6;
  const int per_wave = wave_lds_common_bytes(ix.stride);
  const WaveLds L = carve_wave_lds(smem + (size_t)wv * per_wave, ix.stride, 0, true);
  const int64_t ntq = A.plan[P_NTQ];
  for (int64_t tq = (int64_t)blockIdx.x * 4 + wv; tq < ntq; tq += (int64_t)gridDim.x * 4) {
    const GemmGroup grp = A.groups[A.tq_group[tq]];
    const int qrow = A.gq[tq];
    float q2 = 0.f;
    for (int i = lane; i < ix.stride; i += 64) {
      const float v = (i < ix.d) ? A.queries[(int64_t)qrow * ix.d + i] : 0.f;
      L.qv[i] = v;
      q2 = fmaf(v, v, q2);
    }
    for (int o = 32; o > 0; o >>= 1) q2 += __shfl_xor(q2, o);
    // the slices' lists: one key per lane and slice
    u64 kk[kMaxChunks];
    int total = 0;
    float cut = kInf;
#pragma unroll
    for (int c = 0; c < kMaxChunks; c++) {
      kk[c] = ~0ull;
      if (c < grp.nch) {
        const int64_t slot = tq * kMaxChunks + c;
        const int n = A.cand_cnt[slot];
        if (lane < n) kk[c] = A.cand_key[slot * kCandCap + lane];
        total += n;
        cut = fminf(cut, A.cand_cut[slot]);
      }
    }
    if (total > kSelect) {  // quick-select the key with exactly kSelect smaller keys
      u64 lo = 0, hi = ~0ull, pk = 0;
      for (;;) {
        bool found = false;
#pragma unroll
        for (int c = 0; c < kMaxChunks; c++) {
          const u64 m = ballot64(kk[c] >= lo && kk[c] < hi);
          if (!found && m) {
            pk = rdlane64(kk[c], ctz64(m));
            found = true;
          }
        }
        int below = 0;
#pragma unroll
        for (int c = 0; c < kMaxChunks; c++) below += popc64(ballot64(kk[c] < pk));
        if (below == kSelect) break;
        if (below < kSelect) lo = pk + 1;
        else hi = pk;
      }
      cut = fminf(cut, funkey((uint32_t)(pk >> 32)));  // every listed key that is not taken scores >= the pivot
#pragma unroll
      for (int c = 0; c < kMaxChunks; c++)
        if (!(kk[c] < pk)) kk[c] = ~0ull;
    }
    int cnt = 0;
#pragma unroll
    for (int c = 0; c < kMaxChunks; c++) {
      const u64 m = ballot64(kk[c] != ~0ull);
      if (kk[c] != ~0ull) L.cand_id[cnt + popc64(m & lanemask_lt())] = (int32_t)(uint32_t)kk[c];  // window position
      cnt += popc64(m);
    }
    WAVE_SYNC();
    int rid = 0;
    if (lane < cnt) rid = ix.fi_sorted[grp.a + L.cand_id[lane]];
    WAVE_SYNC();
    L.cand_id[lane] = rid;
    WAVE_SYNC();
    const float dist = wave_distances<METRIC>(ix, L.cand_id, L.cand_dist, L.qv, cnt, 0);
    const u64 key = (lane < cnt) ? (((u64)fkey(dist) << 32) | (uint32_t)rid) : ~0ull;
    int rank = 0;
    for (int l = 0; l < cnt; l++) {
      const u64 kl = rdlane64(key, l);
      rank += (kl < key || (kl == key && l < lane)) ? 1 : 0;
    }
    const int ti = qrow;  // stand-alone PrefilterIndex: one task slot per query
    if (lane < cnt && rank < A.k) A.out_key[(size_t)ti * A.k + rank] = key;
    // proof: every point that was not taken has score >= cut, and |score - exact distance| <= E.
    // E: dropped products q1 p3 + q3 p1 + q2 p2 + ... <= 3.02 * 2^-16 |q||p| (Cauchy-Schwarz over the columns), fp32
    // accumulation of 3 d products (generous factor 8), fp32 norms and the reference's own rounding.
    const float pmax = __uint_as_float(*A.pnorm2_max_bits);
    const float cerr = 3.02f * 1.52587890625e-5f + 8.f * (float)(3 * ix.d + 8) * 5.9604645e-8f;
    const float E = (METRIC == 1) ? cerr * sqrtf(q2 * pmax) : 2.f * cerr * (q2 + pmax);
    if (METRIC != 1) cut += q2;  // the L2 scores leave |q|^2 out
    const int kk_out = cnt < A.k ? cnt : A.k;
    float dk = -3.402823466e+38f;  // k-th exact distance (the worst one that is returned)
    {
      const u64 hit = ballot64(lane < cnt && rank == kk_out - 1);
      if (hit) dk = __shfl(dist, ctz64(hit));
    }
    const bool proven = (cut == kInf) || (dk + E < cut - E);
    if (lane == 0) {
      A.out_cnt[ti] = kk_out;
      if (!proven) {
        A.brute_list[atomicAdd(A.brute_count, 1)] = ti;
        atomicAdd(&ctr->gemm_unproven, 1ull);
      }
    }
    WAVE_SYNC();
  }
}

// ------------------------------------------------------------------------------------------------
static thread_local const char *g_gerr = "";
const char *gemm_launch_last_error() { return g_gerr; }
static int gcheck(hipError_t e) {
  if (e != hipSuccess) {
    g_gerr = hipGetErrorString(e);
    return 1;
  }
  return 0;
}

int launch_point_norms(const IndexView &ix, float *norm2, unsigned int *max_bits, void *stream) {
  if (ix.n <= 0) return 0;
  const int wpb = 4;
  hipLaunchKernelGGL(k_point_norms, dim3((unsigned)((ix.n + wpb - 1) / wpb)), dim3(64 * wpb), 0, (hipStream_t)stream, ix, norm2,
                     max_bits);
  return gcheck(hipGetLastError());
}

int launch_group_windows(const GemmArgs &a, Counters *ctr, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  const int cap = a.cap_mask + 1;
  hipLaunchKernelGGL(k_group_clear, dim3((cap + 255) / 256), dim3(256), 0, s, a);
  hipLaunchKernelGGL(k_group_insert, dim3((unsigned)((a.nq + 255) / 256)), dim3(256), 0, s, a);
  hipLaunchKernelGGL(k_group_plan, dim3(1), dim3(1024), 0, s, a, ctr);
  hipLaunchKernelGGL(k_group_scatter, dim3((unsigned)((a.nq + 255) / 256)), dim3(256), 0, s, a);
  return gcheck(hipGetLastError());
}

size_t gemm_select_lds_bytes(int stride) { return (size_t)128 * (4 * stride + 16) + (size_t)128 * kCandCap * 8 + 4 * 128 * 4; }

int launch_gemm_select(const GemmArgs &a, int num_cus, void *stream) {
  const size_t lds = gemm_select_lds_bytes(a.ix.stride);
  if (lds > 160 * 1024 || a.ix.stride > 128) {
    g_gerr = "dimension too large for the dense prefilter tile";
    return 1;
  }
  void (*kern)(GemmArgs) = nullptr;
  switch (a.ix.stride) {
    case 16: kern = k_gemm_select<16>; break;
    case 32: kern = k_gemm_select<32>; break;
    case 48: kern = k_gemm_select<48>; break;
    case 64: kern = k_gemm_select<64>; break;
    case 80: kern = k_gemm_select<80>; break;
    case 96: kern = k_gemm_select<96>; break;
    case 112: kern = k_gemm_select<112>; break;
    case 128: kern = k_gemm_select<128>; break;
    default: g_gerr = "row stride is not a multiple of 16 floats"; return 1;
  }
  if (gcheck(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds))) return 1;
  hipLaunchKernelGGL(kern, dim3(num_cus > 0 ? num_cus : 256), dim3(256), lds, (hipStream_t)stream, a);
  return gcheck(hipGetLastError());
}

int launch_rerank(const GemmArgs &a, Counters *ctr, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  const int blocks = (int)std::min<int64_t>(4096, (a.nq + 3) / 4);
  const size_t lds = (size_t)4 * (((a.ix.stride * 4 + 15) & ~15) + 64 * 8 + 64 * 4 + 64 * 4);
  if (a.ix.metric == 1) hipLaunchKernelGGL(k_rerank<1>, dim3(blocks), dim3(256), lds, s, a, ctr);
  else hipLaunchKernelGGL(k_rerank<0>, dim3(blocks), dim3(256), lds, s, a, ctr);
  return gcheck(hipGetLastError());
}

}  // namespace wann
