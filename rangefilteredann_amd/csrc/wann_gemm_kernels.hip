// wann_gemm_kernels.hip -- the dense prefilter path: when many queries share one label window
// (PrefilterIndex::batch_search, src/prefiltering.h:124-204 -- e.g. the adversarial dataset, where 99
// queries share each 10 000-point window) the brute-force scan is a true Q x N contraction and runs
// on the matrix cores.  Everything is planned and run on the device; the host only enqueues:
//
//   k_group_clear / k_group_insert / k_group_plan / k_group_scatter
//                    group the batch's queries by window (open-addressing table over (a, b)), lay out the groups'
//                    query lists, score matrices and tiles, hand the ungrouped queries to the exact scan
//   k_gemm_scores    per (window group, 128-query tile, 2 048-position slice): S = Q . P^T on
//                    v_mfma_f32_32x32x16_bf16 with both operands split into two bf16 terms (q = q1 + q2 + ...;
//                    three products q1 p1 + q1 p2 + q2 p1, fp32 accumulate: 2^-16 relative instead of bf16's 2^-8, at
//                    3/16 of the fp32-MFMA cost), scores -q.p (MIPS) or |p|^2 - 2 q.p (L2; |q|^2 joins later).
//                    The scores never reach memory: every lane owns 64 of them per step (one query, 64 window
//                    positions) and keeps their four smallest in registers (min / max insertion, no branches; the
//                    position travels in the six low mantissa bits); one 16-byte store per lane and step leaves
//   select_scores    (first half of k_rerank) per query: the 32 best of its blocks' (three smallest) entries; the fourth smallest of every block
//                    bounds what the block did not hand over
//   k_rerank         per query: exact reference-order distances of those 32 candidates, ordered by
//                    (dist, id), first k; plus a proof that no unselected point can belong to the
//                    top k (score error bound); queries that cannot be proven fall back to
//                    the exact scan kernel k_brute
//
// The MFMA scores only SELECT candidates (the reference sums in another order); every returned
// distance is computed by the reference-order routines.
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
#ifdef WANN_GEMM_PROF  // dev tool (make PROFILE=1): cycles per phase of k_gemm_scores, summed over waves into GemmArgs::prof
#define GPROF_T(v) const unsigned long long v = __builtin_readcyclecounter();
#define GPROF_ADD(i, a, b) prof_acc[i] += (b) - (a);
#else
#define GPROF_T(v)
#define GPROF_ADD(i, a, b)
#endif

constexpr float kHuge = 3.0e38f, kHugeTest = 1.0e38f;  // stands for 'no score' where the bits must stay finite

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
    A.slot_group[i] = -1;  // (k_group_plan only writes the slots that become groups)
  }
  if (i < P_INTS) A.plan[i] = 0;
  if (i == 0) *A.score_used = 0;
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
  bool opened = false;
  for (;;) {
    const unsigned long long old = atomicCAS(&A.slot_key[pos], kEmptySlot, key);
    opened = old == kEmptySlot;  // this thread opened the slot
    if (old == kEmptySlot || old == key) break;
    pos = (pos + 1) & (uint32_t)A.cap_mask;
  }
  {  // the opened slots go on the list: one counter update per wave (a batch of distinct windows opens one per query)
    const unsigned long long om = __ballot(opened);
    if (om) {
      const int lane = threadIdx.x & 63, leader = __builtin_ctzll(om);
      int base = 0;
      if (lane == leader) base = atomicAdd(&A.plan[P_NSLOTS], __builtin_popcountll(om));
      base = __shfl(base, leader);
      if (opened) A.slot_list[base + __builtin_popcountll(om & ((1ull << lane) - 1ull))] = (int32_t)pos;
    }
  }
  A.q_slot[q] = (int32_t)pos;
  const int rank = atomicAdd(&A.slot_count[pos], 1);
  A.q_rank[q] = rank;
  // (the query that makes a slot a group says so: k_group_plan has nothing to do for a batch of distinct windows)
  if (rank == kGroupMinQueries - 1 && t.b - t.a >= kGroupMinWindow) A.plan[P_ANY] = 1;
}

// exclusive prefix of v over the 1024 threads of the workgroup (+ the total): shuffles inside a wave, one pass over the 16
// wave totals.  `wsum` is 16 entries of LDS; two barriers.
template <typename T>
__device__ __forceinline__ T block_excl_scan(T v, T *wsum, T &total) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  T inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const T u = __shfl_up(inc, o);
    if (lane >= o) inc += u;
  }
  __syncthreads();  // (wsum may still be read from the previous scan)
  if (lane == 63) wsum[wv] = inc;
  __syncthreads();
  T off = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) {
    const T w = wsum[i];
    off += (i < wv) ? w : (T)0;
    tot += w;
  }
  total = tot;
  return off + inc - v;
}

// one workgroup: every occupied slot becomes a group (or is left to the exact scan), with its share of the query
// list, of the score buffer and of the tile numbers -- prefix sums over the slots, 1024 at a time (a hundred threads adding to
// the same three counters cost 20 us of serialised atomics)
__global__ __launch_bounds__(1024) void k_group_plan(GemmArgs A, Counters *ctr) {
  __shared__ unsigned long long wsum64[16];
  __shared__ int wsum32[16];
  const int tid = threadIdx.x;
  if (tid == 0) *A.brute_count = 0;  // k_group_scatter rebuilds the exact-scan list
  const int nslots = A.plan[P_NSLOTS];
  unsigned long long used = 0;
  int ngroups = 0, ntq = 0, ntiles = 0;
  // (a batch of distinct windows -- thousands of slots, no group: k_group_insert would have said so)
  if (A.plan[P_ANY] == 0) {
    if (tid == 0) ctr->gemm_queries = 0;
    return;  // (the plan's counts are zero and every slot's group is -1 already: k_group_clear)
  }
  for (int i0 = 0; i0 < nslots; i0 += blockDim.x) {
    const int i = i0 + tid;
    int pos = 0, qc = 0;
    int64_t a = 0, b = 0, w = 0;
    bool eligible = false;
    if (i < nslots) {
      pos = A.slot_list[i];
      const unsigned long long key = A.slot_key[pos];
      qc = A.slot_count[pos];
      a = (int64_t)(key >> 32);
      b = (int64_t)(key & 0xffffffffull);
      w = b - a;
      eligible = qc >= kGroupMinQueries && w >= kGroupMinWindow;
    }
    // entries: per query and 128-position step two blocks (one per half wave) of four floats
    const unsigned long long need = eligible ? (unsigned long long)qc * (unsigned long long)((w + 127) >> 7) * 8ull : 0ull;
    unsigned long long need_total;
    const unsigned long long soff = used + block_excl_scan(need, wsum64, need_total);
    const bool fits = eligible && soff + need <= (unsigned long long)A.score_cap;
    const int nqt = (qc + 127) >> 7, nch = (int)((w + kGemmPointChunk - 1) / kGemmPointChunk);
    int g_total, q_total, t_total;
    const int g = ngroups + block_excl_scan(fits ? 1 : 0, wsum32, g_total);
    const int qoff = ntq + block_excl_scan(fits ? qc : 0, wsum32, q_total);
    const int tile0 = ntiles + block_excl_scan(fits ? nqt * nch : 0, wsum32, t_total);
    if (fits) {
      GemmGroup G;
      G.a = a;
      G.b = b;
      G.soff = (int64_t)soff;
      G.qoff = qoff;
      G.qcount = qc;
      G.nqt = nqt;
      G.nch = nch;
      G.tile0 = tile0;
      G.pad = 0;
      A.groups[g] = G;
      for (int t = 0; t < nqt * nch; t++) A.tile_group[tile0 + t] = g;
    }
    if (fits) A.slot_group[pos] = g;
    used += need_total;
    ngroups += g_total;
    ntq += q_total;
    ntiles += t_total;
  }
  if (tid == 0) {
    A.plan[P_NGROUPS] = ngroups;
    A.plan[P_NTQ] = ntq;
    A.plan[P_NTILES] = ntiles;
    *A.score_used = used;
    ctr->gemm_queries = (unsigned long long)ntq;
  }
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
}

// ------------------------------------------------------------------------------------------------
// GEMM
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

// x into the sorted m1 <= m2 <= m3 <= m4 (the largest drops out): m_i' = x clamped to [m_(i-1), m_i], four independent
// instructions.  (Written as instructions: through the builtins the compiler first canonicalises x -- it is made of integer
// operations, a signalling NaN for all it knows -- with a fifth one; the scores are finite.)
__device__ __forceinline__ void insert4(float &m1, float &m2, float &m3, float &m4, float x) {
  asm volatile("v_med3_f32 %0, %1, %2, %0" : "+v"(m4) : "v"(m3), "v"(x));
  asm volatile("v_med3_f32 %0, %1, %2, %0" : "+v"(m3) : "v"(m2), "v"(x));
  asm volatile("v_med3_f32 %0, %1, %2, %0" : "+v"(m2) : "v"(m1), "v"(x));
  asm volatile("v_min_f32 %0, %0, %1" : "+v"(m1) : "v"(x));
}

__device__ __forceinline__ void insert4_chain(float &m1, float &m2, float &m3, float &m4, float x) {  // the same, seven dependent ones
  float a = fminf(m1, x);
  x = fmaxf(m1, x);
  m1 = a;
  a = fminf(m2, x);
  x = fmaxf(m2, x);
  m2 = a;
  a = fminf(m3, x);
  x = fmaxf(m3, x);
  m3 = a;
  m4 = fminf(m4, x);
}

// One workgroup (4 waves, one per SIMD) per tile = (group, 128 queries, one slice of the window); tiles are taken
// round-robin by a grid of one workgroup per CU.  Per step the workgroup stages 128 points in the LDS as [hi | lo] bf16
// rows; every wave owns 32 query rows (A operand: bf16 pairs in registers for the whole tile) and scores them against
// all 128 points = 1 x 4 MFMA tiles; a score row of 128 floats leaves as four 128-byte stores.
template <int STRIDE>  // padded row length in floats: a multiple of 16, <= 128
__global__ __launch_bounds__(256, 2) void k_gemm_scores(GemmArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const IndexView &ix = A.ix;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  constexpr int S = STRIDE / 16;       // MFMA k-steps per product term
  constexpr int RB = 4 * STRIDE + 16;  // bytes per staged point: hi row, lo row, 16 B so that 8 rows cover all banks
  unsigned char *Ps = smem;                                    // [128][RB]
  float *base = reinterpret_cast<float *>(smem + 128 * RB);    // [128] per staged point: |p|^2 / 0
  int *rid = reinterpret_cast<int *>(base + 128);              // [128] point rows of the block being fetched
  constexpr int s4 = STRIDE >> 2;
  constexpr int nit = s4 >> 1;  // 128 rows x s4 float4 / 256 threads (s4 is even)
  constexpr int nx = s4 >> 2;   // staging: four threads per point row (64 B contiguous), 64 rows per pass, two passes
  const int half = lane >> 5, col = lane & 31;
  const bool mips = ix.metric == 1;
  const float scale = mips ? -1.f : -2.f;
  const int ntiles = A.plan[P_NTILES];
#ifdef WANN_GEMM_PROF
  unsigned long long prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long tk0 = __builtin_readcyclecounter();
#endif

  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const GemmGroup grp = A.groups[A.tile_group[t]];
    const int tl = t - grp.tile0, ch = tl / grp.nqt, q0 = (tl - ch * grp.nqt) << 7;
    const int64_t w = grp.b - grp.a, wlast = w - 1;
    const int64_t p_begin = (int64_t)ch * kGemmPointChunk;
    const int64_t p_end = (p_begin + kGemmPointChunk < w) ? (p_begin + kGemmPointChunk) : w;
    __syncthreads();  // the previous tile is done with the staging area
    // (row numbers fetched ahead are clamped to THIS tile's last position: the block behind a tile's end belongs to another
    // workgroup -- fetching it, rows and all, was 6 % of the kernel's traffic)
    const int64_t tlast = p_end - 1;
    if (tid < 128) rid[tid] = ix.fi_sorted[grp.a + min(p_begin + tid, wlast)];
    // A operand: row 32 wv + col, columns 16 s + 8 half + (0..7).  The query tile passes through the LDS (where
    // the points will be staged) so that the global loads are coalesced; loads are unconditional (clamped indices,
    // select afterwards).
    u32x4 ah[S], al[S];
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
      for (int s = 0; s < S; s++) {
        const float *qp = Qs + (32 * wv + col) * DP + 16 * s + 8 * half;
        const f32x4 v0 = *reinterpret_cast<const f32x4 *>(qp), v1 = *reinterpret_cast<const f32x4 *>(qp + 4);
        uint32_t h, l;
        split2(v0[0], v0[1], h, l); ah[s][0] = h; al[s][0] = l;
        split2(v0[2], v0[3], h, l); ah[s][1] = h; al[s][1] = l;
        split2(v1[0], v1[1], h, l); ah[s][2] = h; al[s][2] = l;
        split2(v1[2], v1[3], h, l); ah[s][3] = h; al[s][3] = l;
      }
    }
    __syncthreads();
    // The MFMA tile has the points as rows and the queries as columns: this lane holds query 32 wv + col, and register
    // reg of tile j the window position 32 j + (reg & 3) + 8 (reg >> 2) + 4 half: 64 positions of one query per step.
    const int myrow = q0 + 32 * wv + col;
    const bool live = myrow < grp.qcount;
    // this lane's entries: [query][step of the window][half] x 4 floats
    const int64_t nsteps = (w + 127) >> 7;
    f32x4 *erow = reinterpret_cast<f32x4 *>(A.scores + grp.soff) + ((int64_t)(live ? myrow : q0) * nsteps + (p_begin >> 7)) * 2 + half;
    // The next point block travels HBM -> registers while the MFMA loop of the current one runs (one wave per SIMD:
    // the 512-register budget is all ours), and registers -> bf16 pairs -> LDS after the barrier.  Its row numbers
    // were put in the LDS one step earlier, so no load depends on another load.
    f32x4 pre[nit];
    float pre_n = 0.f;
    int pre_rid = 0;
#define WANN_FETCH(C0)                                                                                     \
  {                                                                                                        \
    _Pragma("unroll") for (int p = 0; p < 2; p++) {                                                        \
      const float *src = ix.points + (int64_t)rid[64 * p + (tid >> 2)] * STRIDE + 4 * (tid & 3);           \
      _Pragma("unroll") for (int x = 0; x < nx; x++) pre[p * nx + x] = *reinterpret_cast<const f32x4 *>(src + 16 * x); \
    }                                                                                                      \
    if (tid < 128) {                                                                                       \
      if (!mips) pre_n = A.pnorm2[rid[tid]];  /* (inner product: no |p|^2 -- a 4-byte gather costs a 128-byte line per point) */ \
      pre_rid = ix.fi_sorted[grp.a + min((C0) + 128 + tid, tlast)];                                        \
    }                                                                                                      \
  }
    WANN_FETCH(p_begin)
    f32x16 acc[4];
    for (int64_t c0 = p_begin; c0 < p_end; c0 += 128) {
      GPROF_T(t0)
      // (the barrier that ended the previous step: nobody reads Ps / base / rid any more)
#pragma unroll
      for (int p = 0; p < 2; p++) {
        unsigned char *dst = Ps + (64 * p + (tid >> 2)) * RB + 8 * (tid & 3);
#pragma unroll
        for (int x = 0; x < nx; x++) {
          const f32x4 v = pre[p * nx + x];
          uint32_t h0, l0, h1, l1;
          split2(v[0], v[1], h0, l0);
          split2(v[2], v[3], h1, l1);
          *reinterpret_cast<uint2 *>(dst + 32 * x) = make_uint2(h0, h1);
          *reinterpret_cast<uint2 *>(dst + 2 * STRIDE + 32 * x) = make_uint2(l0, l1);
        }
      }
      if (tid < 128) {
        base[tid] = (c0 + tid < p_end) ? (mips ? 0.f : pre_n) : kHuge;  // positions beyond the window never win
        rid[tid] = pre_rid;
      }
      GPROF_T(t1)
      __syncthreads();
      GPROF_T(t2)
      WANN_FETCH(c0 + 128)  // unconditional (row numbers are clamped): a conditional fetch would make the compiler wait for it here
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[j][r] = 0.f;
      // B operand: eight ds_read_b128 per k-step feed twelve MFMAs; two workgroups share a CU, so the other wave of the
      // SIMD fills the gaps (its MFMAs run under this wave's conversions and insertions, and the other way round)
      const unsigned char *pb = Ps + col * RB + 16 * half;
#pragma unroll
      for (int s = 0; s < S; s++) {
        const bf16x8 a_hi = __builtin_bit_cast(bf16x8, ah[s]), a_lo = __builtin_bit_cast(bf16x8, al[s]);
        bf16x8 bh[4], bl[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          bh[j] = *reinterpret_cast<const bf16x8 *>(pb + j * 32 * RB + 32 * s);
          bl[j] = *reinterpret_cast<const bf16x8 *>(pb + j * 32 * RB + 2 * STRIDE + 32 * s);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl[j], a_hi, acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; j++) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[j], a_lo, acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; j++) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[j], a_hi, acc[j], 0, 0, 0);
      }
      GPROF_T(t3)
      // the four smallest of this lane's 64 scores, sorted; low six mantissa bits = 16 j + reg (which position)
      float m1 = kHuge, m2 = kHuge, m3 = kHuge, m4 = kHuge;
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int g = 0; g < 4; g++) {
          const f32x4 b4 = *reinterpret_cast<const f32x4 *>(base + 32 * j + 8 * g + 4 * half);
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const float sc = fmaf(scale, acc[j][4 * g + r], b4[r]);
            const float x = __uint_as_float((__float_as_uint(sc) & ~63u) | (uint32_t)(16 * j + 4 * g + r));
            if constexpr (STRIDE < 128) insert4(m1, m2, m3, m4, x);
            else insert4_chain(m1, m2, m3, m4, x);  // (at 128 floats per row the four-instruction form does not fit 256 registers)
          }
        }
      if (live) erow[(c0 - p_begin) >> 6] = f32x4{m1, m2, m3, m4};
      GPROF_T(t4)
      __syncthreads();  // every wave is done with Ps / base / rid
      GPROF_T(t5)
      GPROF_ADD(0, t0, t1) GPROF_ADD(1, t1, t2) GPROF_ADD(2, t2, t3) GPROF_ADD(3, t3, t4) GPROF_ADD(4, t4, t5)
    }
  }
#ifdef WANN_GEMM_PROF
  prof_acc[5] = __builtin_readcyclecounter() - tk0;
  if (lane == 0)
    for (int i = 0; i < 8; i++) atomicAdd(A.prof + i, prof_acc[i]);
#endif
#undef WANN_FETCH
}

// The same kernel for rows of 129 .. 512 floats (RedCaps: d = 512): the dimension is walked in SLABS slabs of 128 floats.
// The A operand -- this wave's 32 query rows, split into bf16 pairs -- stays in registers for ALL slabs (8 SLABS k-steps:
// 256 registers at 512 floats, hence one workgroup per CU and up to 512 registers per wave), the points are staged slab
// by slab through the same LDS area and fetch pipeline, the accumulators run across the slabs of a step, and everything
// after the MFMAs (selection network, hand-over format) is the narrow kernel's.  Columns beyond the row stride (a last,
// partial slab) are zero on both sides.
template <int SLABS>
__global__ __launch_bounds__(256, 1) void k_gemm_scores_wide(GemmArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const IndexView &ix = A.ix;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  constexpr int W = 128;             // slab width in floats
  constexpr int S = W / 16;          // MFMA k-steps per slab
  constexpr int RB = 4 * W + 16;     // bytes per staged point and slab
  unsigned char *Ps = smem;                                    // [128][RB]
  float *base = reinterpret_cast<float *>(smem + 128 * RB);    // [128] per staged point: |p|^2 / 0
  int *rid = reinterpret_cast<int *>(base + 128);              // [128] point rows of the step being fetched (+ a second [128], see below)
  // four slabs: the low halves of the LAST slab's A operand live in the LDS (8 KiB per wave; a lane reads its own 16 bytes):
  // 32 registers that operands, accumulators and the block in flight do not have
  constexpr int SR = SLABS == 4 ? 3 : SLABS;  // slabs whose low halves stay in registers
  u32x4 *const alds = reinterpret_cast<u32x4 *>(smem + 128 * RB + 3 * 128 * 4) + wv * (S * 64) + lane;
  constexpr int s4 = W >> 2, nit = s4 >> 1, nx = s4 >> 2;
  const int half = lane >> 5, col = lane & 31;
  const bool mips = ix.metric == 1;
  const float scale = mips ? -1.f : -2.f;
  const int ntiles = A.plan[P_NTILES];
  const int stride = ix.stride;

  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const GemmGroup grp = A.groups[A.tile_group[t]];
    const int tl = t - grp.tile0, ch = tl / grp.nqt, q0 = (tl - ch * grp.nqt) << 7;
    const int64_t w = grp.b - grp.a, wlast = w - 1;
    const int64_t p_begin = (int64_t)ch * kGemmPointChunk;
    const int64_t p_end = (p_begin + kGemmPointChunk < w) ? (p_begin + kGemmPointChunk) : w;
    __syncthreads();  // the previous tile is done with the staging area
    if (tid < 128) rid[tid] = ix.fi_sorted[grp.a + min(p_begin + tid, wlast)];
    // A operand, slab by slab through the LDS (coalesced global loads): row 32 wv + col, columns 128 sl + 16 s + 8 half + (0..7)
    u32x4 ah[S * SLABS], al[S * SR];
    {
      constexpr int DP = W + 4;
      float *Qs = reinterpret_cast<float *>(Ps);
      const int dlast = ix.d - 1;
#pragma unroll
      for (int sl = 0; sl < SLABS; sl++) {
        if (sl) __syncthreads();
#pragma unroll 2
        for (int it = 0; it < nit; it++) {
          const int idx = tid + it * 256;
          const int r = idx / s4, c = W * sl + (idx - r * s4) * 4;
          const bool live = q0 + r < grp.qcount;
          const float *src = A.queries + (int64_t)A.gq[grp.qoff + (live ? q0 + r : grp.qcount - 1)] * ix.d;
          f32x4 v;
          v[0] = src[min(c + 0, dlast)]; v[1] = src[min(c + 1, dlast)]; v[2] = src[min(c + 2, dlast)]; v[3] = src[min(c + 3, dlast)];
#pragma unroll
          for (int e = 0; e < 4; e++) v[e] = (live && c + e < ix.d) ? v[e] : 0.f;
          *reinterpret_cast<f32x4 *>(Qs + r * DP + (c - W * sl)) = v;
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < S; s++) {
          const float *qp = Qs + (32 * wv + col) * DP + 16 * s + 8 * half;
          const f32x4 v0 = *reinterpret_cast<const f32x4 *>(qp), v1 = *reinterpret_cast<const f32x4 *>(qp + 4);
          uint32_t h, l;
          u32x4 lo4;
          split2(v0[0], v0[1], h, l); ah[S * sl + s][0] = h; lo4[0] = l;
          split2(v0[2], v0[3], h, l); ah[S * sl + s][1] = h; lo4[1] = l;
          split2(v1[0], v1[1], h, l); ah[S * sl + s][2] = h; lo4[2] = l;
          split2(v1[2], v1[3], h, l); ah[S * sl + s][3] = h; lo4[3] = l;
          if (sl < SR) al[S * sl + s] = lo4;
          else alds[s * 64] = lo4;
        }
      }
    }
    __syncthreads();
    const int myrow = q0 + 32 * wv + col;
    const bool live = myrow < grp.qcount;
    const int64_t nsteps = (w + 127) >> 7;
    f32x4 *erow = reinterpret_cast<f32x4 *>(A.scores + grp.soff) + ((int64_t)(live ? myrow : q0) * nsteps + (p_begin >> 7)) * 2 + half;
    // fetch pipeline: the next (step, slab) travels HBM -> registers during the MFMAs of the current one.  `rid` holds the
    // rows of the step being fetched; it moves on to the next step when a step's LAST slab is staged.
    f32x4 pre[nit];
    float pre_n = 0.f;
    int pre_rid = 0;
#define WANN_FETCHW(C0, SL, NEWSTEP)                                                                       \
  {                                                                                                        \
    _Pragma("unroll") for (int p = 0; p < 2; p++) {                                                        \
      const float *src = ix.points + (int64_t)rid[64 * p + (tid >> 2)] * stride;                           \
      _Pragma("unroll") for (int x = 0; x < nx; x++) {                                                     \
        const int cf = W * (SL) + 4 * (tid & 3) + 16 * x;                                                  \
        pre[p * nx + x] = *reinterpret_cast<const f32x4 *>(src + min(cf, stride - 4));                     \
      }                                                                                                    \
    }                                                                                                      \
    if ((NEWSTEP) && tid < 128) {                                                                          \
      if (!mips) pre_n = A.pnorm2[rid[tid]];                                                               \
      pre_rid = ix.fi_sorted[grp.a + min((C0) + 128 + tid, wlast)];                                        \
    }                                                                                                      \
  }
    // (four slabs: the A operand alone is 256 registers -- the fetch is then NOT overlapped with the MFMAs: the 64 registers
    // of a block in flight do not fit beside operands and accumulators)
    constexpr bool PIPE = SLABS < 4;
    if (PIPE) WANN_FETCHW(p_begin, 0, true)
    f32x16 acc[4];
    // Not pipelined (four slabs): every slab of a step reads the step's rows from `rid` while it stages, and the only barrier
    // between the last slab's reads and the hand-over of the next step's rows would be the one that ENDED the slab before --
    // a wave that lags by one gather would stage the next step's points for this step.  The rows therefore alternate between
    // two arrays by step parity: the last slab writes the array nobody reads until the step-ending barrier has passed.
    int *rid_cur = rid, *rid_nxt = PIPE ? rid : rid + 128;
    for (int64_t c0 = p_begin; c0 < p_end; c0 += 128) {
#pragma unroll
      for (int sl = 0; sl < SLABS; sl++) {
        // (the barrier that ended the previous slab: nobody reads Ps / base any more)
        if (!PIPE && sl == 0 && tid < 128) {  // (rid_cur = this step's rows; pre_rid = the next step's, published at the last slab)
          if (!mips) pre_n = A.pnorm2[rid_cur[tid]];
          pre_rid = ix.fi_sorted[grp.a + min(c0 + 128 + tid, wlast)];
        }
#pragma unroll
        for (int p = 0; p < 2; p++) {
          if (!PIPE) {  // fetch and stage half a slab at a time: 32 registers in flight instead of 64
            const float *src = ix.points + (int64_t)rid_cur[64 * p + (tid >> 2)] * stride;
#pragma unroll
            for (int x = 0; x < nx; x++) pre[p * nx + x] = *reinterpret_cast<const f32x4 *>(src + min(W * sl + 4 * (tid & 3) + 16 * x, stride - 4));
          }
          unsigned char *dst = Ps + (64 * p + (tid >> 2)) * RB + 8 * (tid & 3);
#pragma unroll
          for (int x = 0; x < nx; x++) {
            f32x4 v = pre[p * nx + x];
            if (W * sl + 4 * (tid & 3) + 16 * x >= stride) v = f32x4{0.f, 0.f, 0.f, 0.f};  // beyond the row: zero
            uint32_t h0, l0, h1, l1;
            split2(v[0], v[1], h0, l0);
            split2(v[2], v[3], h1, l1);
            *reinterpret_cast<uint2 *>(dst + 32 * x) = make_uint2(h0, h1);
            *reinterpret_cast<uint2 *>(dst + 2 * W + 32 * x) = make_uint2(l0, l1);
          }
        }
        if (tid < 128) {
          if (sl == 0) base[tid] = (c0 + tid < p_end) ? (mips ? 0.f : pre_n) : kHuge;  // positions beyond the window never win
          if (sl == SLABS - 1) rid_nxt[tid] = pre_rid;
        }
        __syncthreads();
        if (PIPE) {
          if (sl + 1 < SLABS) WANN_FETCHW(c0, sl + 1, false)
          else WANN_FETCHW(c0 + 128, 0, true)  // unconditional (row numbers are clamped)
        }
        if (sl == 0) {
#pragma unroll
          for (int j = 0; j < 4; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[j][r] = 0.f;
        }
        const unsigned char *pb = Ps + col * RB + 16 * half;
#pragma unroll
        for (int s = 0; s < S; s++) {
          const bf16x8 a_hi = __builtin_bit_cast(bf16x8, ah[S * sl + s]);
          const bf16x8 a_lo = __builtin_bit_cast(bf16x8, sl < SR ? al[S * sl + s] : alds[s * 64]);
          // (one or two point tiles' operands at a time: registers are what this kernel is short of)
          constexpr int JB = SLABS < 4 ? 2 : 1;
#pragma unroll
          for (int j0 = 0; j0 < 4; j0 += JB) {
            bf16x8 bh[JB], bl[JB];
#pragma unroll
            for (int j = 0; j < JB; j++) {
              bh[j] = *reinterpret_cast<const bf16x8 *>(pb + (j0 + j) * 32 * RB + 32 * s);
              bl[j] = *reinterpret_cast<const bf16x8 *>(pb + (j0 + j) * 32 * RB + 2 * W + 32 * s);
            }
#pragma unroll
            for (int j = 0; j < JB; j++) acc[j0 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl[j], a_hi, acc[j0 + j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < JB; j++) acc[j0 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[j], a_lo, acc[j0 + j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < JB; j++) acc[j0 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[j], a_hi, acc[j0 + j], 0, 0, 0);
          }
        }
        if (sl + 1 < SLABS) __syncthreads();  // every wave is done with this slab's rows
      }
      // the four smallest of this lane's 64 scores, sorted; low six mantissa bits = 16 j + reg (which position)
      float m1 = kHuge, m2 = kHuge, m3 = kHuge, m4 = kHuge;
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int g = 0; g < 4; g++) {
          const f32x4 b4 = *reinterpret_cast<const f32x4 *>(base + 32 * j + 8 * g + 4 * half);
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const float sc = fmaf(scale, acc[j][4 * g + r], b4[r]);
            const float x = __uint_as_float((__float_as_uint(sc) & ~63u) | (uint32_t)(16 * j + 4 * g + r));
            insert4(m1, m2, m3, m4, x);
          }
        }
      if (live) erow[(c0 - p_begin) >> 6] = f32x4{m1, m2, m3, m4};
      __syncthreads();  // every wave is done with Ps / base / this step's rows
      if (!PIPE) {
        int *const t = rid_cur;
        rid_cur = rid_nxt;
        rid_nxt = t;
      }
    }
  }
#undef WANN_FETCHW
}

// Four slabs (rows of 385 .. 512 floats: RedCaps), fetch overlapped with the MFMAs (round 4).  The A operand alone is 256
// registers there, so k_gemm_scores_wide<4> cannot keep a block in flight in registers and every half slab waited for its HBM
// round trip in full, twice a slab.  Here the points travel HBM -> LDS directly (`global_load_lds_dwordx4`: no registers in
// flight): a 32-KiB raw area R holds ONE half slab (64 points x 128 floats, a wave's 1 KiB per instruction, lane-contiguous),
// and the unit of work is a half slab --
//   wait for R | my 128 bytes of it -> bf16 pairs -> Ps rows of this half | barrier | request the NEXT half slab into R |
//   the 48 MFMAs of this half (two point tiles x eight k-steps x three products)
// -- one barrier per unit: the rows a unit stages were last read by the MFMAs two units back (every wave has passed a barrier
// since), R is requested again only after every wave has read its part.  Row numbers and |p|^2 of the NEXT step are fetched
// in units 4 .. 6 of a step and live in arrays that alternate by step parity (no barrier between a step's selection network
// and the next step's staging).  Arithmetic, tile shapes, selection network and hand-over format are k_gemm_scores_wide's.
__global__ __launch_bounds__(256, 1) void k_gemm_scores_wide4(GemmArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const IndexView &ix = A.ix;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  constexpr int SLABS = 4, W = 128, S = W / 16, RB = 4 * W + 16, SR = 3;
  unsigned char *Ps = smem;                                    // [128][RB]
  float *base = reinterpret_cast<float *>(smem + 128 * RB);    // [2][128] per staged point: |p|^2 / 0, by step parity
  int *rid = reinterpret_cast<int *>(base + 256);              // [2][128] point rows of a step, by step parity
  u32x4 *const alds = reinterpret_cast<u32x4 *>(smem + 128 * RB + 4 * 128 * 4) + wv * (S * 64) + lane;  // low halves of the last slab's A operand
  unsigned char *const R = smem + 128 * RB + 4 * 128 * 4 + 4 * S * 64 * 16;                             // raw half slab
  constexpr int s4 = W >> 2, nit = s4 >> 1, nx = s4 >> 2;
  const int half = lane >> 5, col = lane & 31;
  const bool mips = ix.metric == 1;
  const float scale = mips ? -1.f : -2.f;
  const int ntiles = A.plan[P_NTILES];
  const int stride = ix.stride;

  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const GemmGroup grp = A.groups[A.tile_group[t]];
    const int tl = t - grp.tile0, ch = tl / grp.nqt, q0 = (tl - ch * grp.nqt) << 7;
    const int64_t w = grp.b - grp.a, wlast = w - 1;
    const int64_t p_begin = (int64_t)ch * kGemmPointChunk;
    const int64_t p_end = (p_begin + kGemmPointChunk < w) ? (p_begin + kGemmPointChunk) : w;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (nothing of the previous tile is on its way into R any more)
    __syncthreads();  // the previous tile is done with the staging area
    if (tid < 128) rid[tid] = ix.fi_sorted[grp.a + min(p_begin + tid, wlast)];
    u32x4 ah[S * SLABS], al[S * SR];
    {
      constexpr int DP = W + 4;
      float *Qs = reinterpret_cast<float *>(Ps);
      const int dlast = ix.d - 1;
#pragma unroll
      for (int sl = 0; sl < SLABS; sl++) {
        if (sl) __syncthreads();
#pragma unroll 2
        for (int it = 0; it < nit; it++) {
          const int idx = tid + it * 256;
          const int r = idx / s4, c = W * sl + (idx - r * s4) * 4;
          const bool live = q0 + r < grp.qcount;
          const float *src = A.queries + (int64_t)A.gq[grp.qoff + (live ? q0 + r : grp.qcount - 1)] * ix.d;
          f32x4 v;
          v[0] = src[min(c + 0, dlast)]; v[1] = src[min(c + 1, dlast)]; v[2] = src[min(c + 2, dlast)]; v[3] = src[min(c + 3, dlast)];
#pragma unroll
          for (int e = 0; e < 4; e++) v[e] = (live && c + e < ix.d) ? v[e] : 0.f;
          *reinterpret_cast<f32x4 *>(Qs + r * DP + (c - W * sl)) = v;
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < S; s++) {
          const float *qp = Qs + (32 * wv + col) * DP + 16 * s + 8 * half;
          const f32x4 v0 = *reinterpret_cast<const f32x4 *>(qp), v1 = *reinterpret_cast<const f32x4 *>(qp + 4);
          uint32_t h, l;
          u32x4 lo4;
          split2(v0[0], v0[1], h, l); ah[S * sl + s][0] = h; lo4[0] = l;
          split2(v0[2], v0[3], h, l); ah[S * sl + s][1] = h; lo4[1] = l;
          split2(v1[0], v1[1], h, l); ah[S * sl + s][2] = h; lo4[2] = l;
          split2(v1[2], v1[3], h, l); ah[S * sl + s][3] = h; lo4[3] = l;
          if (sl < SR) al[S * sl + s] = lo4;
          else alds[s * 64] = lo4;
        }
      }
    }
    __syncthreads();
    const int myrow = q0 + 32 * wv + col;
    const bool live = myrow < grp.qcount;
    const int64_t nsteps = (w + 127) >> 7;
    f32x4 *erow = reinterpret_cast<f32x4 *>(A.scores + grp.soff) + ((int64_t)(live ? myrow : q0) * nsteps + (p_begin >> 7)) * 2 + half;
    // half slab (SL, HF) of the step whose rows are ROWS -> R: thread (row tid >> 2 of the half, 16-byte column group tid & 3)
    // brings eight pieces, 64 bytes apart; piece x of wave wv lands at R + (4 x + wv) KiB + 16 lane
#define WANN_REQUEST(ROWS, SL, HF)                                                                                        \
  {                                                                                                                       \
    const float *src_ = ix.points + (int64_t)(ROWS)[64 * (HF) + (tid >> 2)] * stride;                                     \
    _Pragma("unroll") for (int x = 0; x < nx; x++)                                                                        \
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src_ + min(W * (SL) + 4 * (tid & 3) + 16 * x, stride - 4)), \
                                       (__attribute__((address_space(3))) void *)(R + (4 * x + wv) * 1024), 16, 0, 0);     \
  }
    float pre_n = 0.f;
    int pre_rid = 0;
    if (tid < 128 && !mips) pre_n = A.pnorm2[rid[tid]];
    WANN_REQUEST(rid, 0, 0)
    f32x16 acc[4];
    int par = 0;
    for (int64_t c0 = p_begin; c0 < p_end; c0 += 128, par ^= 1) {
      int *const rid_cur = rid + 128 * par, *const rid_nxt = rid + 128 * (par ^ 1);
      float *const base_cur = base + 128 * par;
#pragma unroll
      for (int sl = 0; sl < SLABS; sl++) {
#pragma unroll
        for (int hf = 0; hf < 2; hf++) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // my pieces of the unit have landed (and the row numbers / norms asked for earlier)
          {
            unsigned char *dst = Ps + (64 * hf + (tid >> 2)) * RB + 8 * (tid & 3);
#pragma unroll
            for (int x = 0; x < nx; x++) {
              f32x4 v = *reinterpret_cast<const f32x4 *>(R + (4 * x + wv) * 1024 + 16 * lane);
              if (W * sl + 4 * (tid & 3) + 16 * x >= stride) v = f32x4{0.f, 0.f, 0.f, 0.f};  // beyond the row: zero
              uint32_t h0, l0, h1, l1;
              split2(v[0], v[1], h0, l0);
              split2(v[2], v[3], h1, l1);
              *reinterpret_cast<uint2 *>(dst + 32 * x) = make_uint2(h0, h1);
              *reinterpret_cast<uint2 *>(dst + 2 * W + 32 * x) = make_uint2(l0, l1);
            }
          }
          if (tid < 128) {
            if (sl == 0 && hf == 0) base_cur[tid] = (c0 + tid < p_end) ? (mips ? 0.f : pre_n) : kHuge;  // positions beyond the window never win
            if (sl == 2 && hf == 0) pre_rid = ix.fi_sorted[grp.a + min(c0 + 128 + tid, wlast)];
            if (sl == 2 && hf == 1) rid_nxt[tid] = pre_rid;
            if (sl == 3 && hf == 0 && !mips) pre_n = A.pnorm2[rid_nxt[tid]];
          }
          __syncthreads();  // the unit is staged; R is free
          if (hf == 0) WANN_REQUEST(rid_cur, sl, 1)
          else if (sl + 1 < SLABS) WANN_REQUEST(rid_cur, sl + 1, 0)
          else if (c0 + 128 < p_end) WANN_REQUEST(rid_nxt, 0, 0)
          if (sl == 0) {
#pragma unroll
            for (int r = 0; r < 16; r++) acc[2 * hf][r] = acc[2 * hf + 1][r] = 0.f;
          }
          const unsigned char *pb = Ps + col * RB + 16 * half;
#pragma unroll
          for (int s = 0; s < S; s++) {
            const bf16x8 a_hi = __builtin_bit_cast(bf16x8, ah[S * sl + s]);
            const bf16x8 a_lo = __builtin_bit_cast(bf16x8, sl < SR ? al[S * sl + s] : alds[s * 64]);
#pragma unroll
            for (int j = 2 * hf; j < 2 * hf + 2; j++) {
              const bf16x8 bh = *reinterpret_cast<const bf16x8 *>(pb + j * 32 * RB + 32 * s);
              const bf16x8 bl = *reinterpret_cast<const bf16x8 *>(pb + j * 32 * RB + 2 * W + 32 * s);
              acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl, a_hi, acc[j], 0, 0, 0);
              acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, a_lo, acc[j], 0, 0, 0);
              acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, a_hi, acc[j], 0, 0, 0);
            }
          }
        }
      }
      // the four smallest of this lane's 64 scores, sorted; low six mantissa bits = 16 j + reg (which position)
      float m1 = kHuge, m2 = kHuge, m3 = kHuge, m4 = kHuge;
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int g = 0; g < 4; g++) {
          const f32x4 b4 = *reinterpret_cast<const f32x4 *>(base_cur + 32 * j + 8 * g + 4 * half);
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const float sc = fmaf(scale, acc[j][4 * g + r], b4[r]);
            const float x = __uint_as_float((__float_as_uint(sc) & ~63u) | (uint32_t)(16 * j + 4 * g + r));
            insert4(m1, m2, m3, m4, x);
          }
        }
      if (live) erow[(c0 - p_begin) >> 6] = f32x4{m1, m2, m3, m4};
    }
#undef WANN_REQUEST
  }
}

// One wave per grouped query.  Its window's blocks each handed over their four smallest scores (sorted, position in
// the low mantissa bits).  The first three of every block are candidates, the fourth bounds everything the block kept
// to itself.  The kSelect best candidates live sorted in lanes 0 .. kSelect-1 (score bits in one register, window
// positions in another); candidates below the current cut are inserted one by one with a ballot + one-lane shift.
// (one wave, one query; result in registers: lane l < filled holds the window-relative position of a selected candidate,
// `cut` / `blk_bound` are the two bounds on everything that was not selected, FLT_MAX = nothing was left out that way)
__device__ __forceinline__ void select_scores(const f32x4 *erow, int64_t nblk, int &sel_pos, int &sel_cnt, float &cut, float &blk_bound) {
  const int lane = lane_id();
  {
    uint32_t top_s = 0xffffffffu, thr = 0xffffffffu;  // 0xffffffff (no float maps to it) = empty slot; thr = lane kSelect-1
    int top_p = 0, filled = 0;
    float bound = kHuge;
    for (int64_t b0 = 0; b0 < nblk; b0 += 64) {
      const int64_t blk = b0 + lane;
      const f32x4 e = (blk < nblk) ? erow[blk] : f32x4{kHuge, kHuge, kHuge, kHuge};
      bound = fminf(bound, e[3]);
      if (b0 == 0) {
        // The list starts as the kSelect smallest of the first 64 blocks' MINIMA, by one bitonic sort across the wave (21
        // exchange steps) instead of ~64 insertions: the threshold is tight from the start (half of the block minima are
        // below it, one candidate in ten of the rest), and every later candidate is tested against it before it costs a
        // serial insertion -- 45 of them per query instead of 120 on a 10 000-point window.
        uint32_t key = (e[0] < kHugeTest) ? fkey(e[0]) : 0xffffffffu;
        const uint32_t ix6 = __float_as_uint(e[0]) & 63u;
        int pos = (int)((lane >> 1) * 128 + 32 * (ix6 >> 4) + 8 * ((ix6 >> 2) & 3) + 4 * (lane & 1) + (ix6 & 3));
#pragma unroll
        for (int k = 2; k <= 64; k <<= 1)
#pragma unroll
          for (int j = k >> 1; j > 0; j >>= 1) {
            const uint32_t ok = (uint32_t)__shfl_xor((int)key, j);
            const int op = __shfl_xor(pos, j);
            const bool take_min = ((lane & j) == 0) == ((lane & k) == 0);
            const bool swap = take_min ? (ok < key) : (ok > key);  // (equal keys stay where they are)
            key = swap ? ok : key;
            pos = swap ? op : pos;
          }
        top_s = (lane < kSelect) ? key : 0xffffffffu;
        top_p = pos;
        filled = popc64(ballot64(lane < kSelect && key != 0xffffffffu));
        thr = (uint32_t)rdlane((int)top_s, kSelect - 1);
      }
#pragma unroll
      for (int c = 0; c < 3; c++) {
        if (c == 0 && b0 == 0) continue;  // (placed above)
        const uint32_t key = fkey(e[c]);
        u64 mask = ballot64(e[c] < kHugeTest && key < thr);
        while (mask) {
          const int src = ctz64(mask);
          mask &= mask - 1;
          const uint32_t ck = (uint32_t)rdlane((int)key, src);
          if (ck < thr) {  // wave-uniform; thr may have dropped since the ballot
            const int p = popc64(ballot64(top_s <= ck));  // top is sorted: a prefix of the lanes
            // block b0 + src: step (b >> 1), half (b & 1); low bits 16 j + 4 g + r -> position 32 j + 8 g + 4 half + r
            const uint32_t ix6 = (uint32_t)rdlane((int)__float_as_uint(e[c]), src) & 63u;
            const int64_t bb = b0 + src;
            const int cp = (int)((bb >> 1) * 128 + 32 * (ix6 >> 4) + 8 * ((ix6 >> 2) & 3) + 4 * (bb & 1) + (ix6 & 3));
            const uint32_t up_s = (uint32_t)__builtin_amdgcn_update_dpp((int)top_s, (int)top_s, 0x138, 0xf, 0xf, false);
            const int up_p = __builtin_amdgcn_update_dpp(top_p, top_p, 0x138, 0xf, 0xf, false);
            if (lane < kSelect) {
              top_s = (lane == p) ? ck : (lane > p ? up_s : top_s);
              top_p = (lane == p) ? cp : (lane > p ? up_p : top_p);
            }
            filled += filled < kSelect;
            thr = (uint32_t)rdlane((int)top_s, kSelect - 1);
          }
        }
      }
    }
    for (int o = 32; o > 0; o >>= 1) bound = fminf(bound, __shfl_xor(bound, o));
    // every position that is not selected scores >= cut: candidates that were dropped or never inserted >= the worst
    // selected one (once the list is full), everything else >= its block's fourth smallest
    sel_pos = top_p;
    sel_cnt = filled;
    cut = (filled == kSelect) ? funkey(thr) : 3.402823466e+38f;
    blk_bound = (bound >= kHugeTest) ? 3.402823466e+38f : bound;
  }
}


// one wave per grouped query: exact distances of the selected candidates, (dist, id) order, proof
template <int METRIC>
__global__ __launch_bounds__(256) void k_rerank(GemmArgs A, Counters *ctr) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const IndexView &ix = A.ix;
  const int lane = lane_id(), wv = threadIdx.x >> 6;
  const int K = A.k;
  const int per_wave = wave_lds_common_bytes(ix.stride) + ((K + 1) & ~1) * 8;
  const WaveLds L = carve_wave_lds(smem + (size_t)wv * per_wave, ix.stride, K, true);
  const int64_t ntq = A.plan[P_NTQ];
  for (int64_t tq = (int64_t)blockIdx.x * 4 + wv; tq < ntq; tq += (int64_t)gridDim.x * 4) {
    const GemmGroup grp = A.groups[A.tq_group[tq]];
    const int qrow = A.gq[tq];
    float q2 = 0.f;  // (the query row is on its way while the selection runs)
    for (int i = lane; i < ix.stride; i += 64) {
      const float v = (i < ix.d) ? A.queries[(int64_t)qrow * ix.d + i] : 0.f;
      L.qv[i] = v;
      q2 = fmaf(v, v, q2);
    }
    const int64_t nblk_sel = ((grp.b - grp.a + 127) >> 7) * 2;
    int sel_pos, cnt;
    float cut_sel, cut_blk;
    select_scores(reinterpret_cast<const f32x4 *>(A.scores + grp.soff) + (int64_t)A.tq_local[tq] * nblk_sel, nblk_sel, sel_pos, cnt, cut_sel, cut_blk);
    for (int o = 32; o > 0; o >>= 1) q2 += __shfl_xor(q2, o);
    int rid = 0;
    if (lane < cnt) rid = ix.fi_sorted[grp.a + sel_pos];
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
    if (lane < cnt && rank < K) A.out_key[(size_t)ti * K + rank] = key;
    // proof: every unselected point has score >= cut, and |score - exact distance| <= E.
    // E: the products the bf16 split drops (q1 p3 + q3 p1 + q2 p2 + ...) <= 3.02 * 2^-16 |q||p| (Cauchy-Schwarz over
    // the columns), fp32 accumulation of 3 d products (A.acc_factor x the rounding adder's worst case), fp32 norms and the
    // reference's own rounding.
    const float pmax = __uint_as_float(*A.pnorm2_max_bits);
    const float cerr = 3.02f * 1.52587890625e-5f + A.acc_factor * (float)(3 * ix.d + 8) * 5.9604645e-8f;
    // + 2^-17 relative for the six mantissa bits that carry the position (|score| <= |q||p| resp. 2 (|q|^2 + |p|^2))
    const float cerr2 = cerr + 7.62939453125e-6f;
    const float E = (METRIC == 1) ? cerr2 * sqrtf(q2 * pmax) : 2.f * cerr2 * (q2 + pmax);
    const int kk = cnt < K ? cnt : K;
    float dk = -3.402823466e+38f;  // k-th exact distance (the worst one that is returned)
    {
      const u64 hit = ballot64(lane < cnt && rank == kk - 1);
      if (hit) dk = __shfl(dist, ctz64(hit));
    }
    // two bounds on what was not selected: candidates that lost against the selected ones (>= the worst selected), and
    // whatever the blocks kept to themselves (>= the smallest fourth entry); FLT_MAX = no such position exists
    const float qoff = (METRIC == 1) ? 0.f : q2;  // the L2 scores leave |q|^2 out
    const bool sel_ok = cut_sel == 3.402823466e+38f || (cnt >= K && dk + E < cut_sel + qoff - E);
    const bool blk_ok = cut_blk == 3.402823466e+38f || (cnt >= K && dk + E < cut_blk + qoff - E);
    bool proven = sel_ok && blk_ok;
    int outn = kk;
    if (!proven && sel_ok && cnt >= K) {
      // Second chance: only some blocks' kept positions could still matter (fourth entry - E <= d_k + E; a fixed set, d_k
      // can only improve).  Score those blocks exactly, 64 positions each, and merge; all other blocks stay proven.
      // (Labels that correlate with the geometry put a query's best points next to each other: the same block.)
      int m = 0, p0;
      m = wave_merge(L.lbeam, m, K, lane < cnt, ((u64)fkey(dist) << 32) | ((u64)(uint32_t)rid << 1), L.cand_key, &p0);
      const int64_t w = grp.b - grp.a, nblk = ((w + 127) >> 7) * 2;
      const f32x4 *erow = reinterpret_cast<const f32x4 *>(A.scores + grp.soff) + (int64_t)A.tq_local[tq] * nblk;
      int scanned = 0;
      bool gave_up = false;
      for (int64_t b0 = 0; b0 < nblk && !gave_up; b0 += 64) {
        const float m4 = (b0 + lane < nblk) ? erow[b0 + lane][3] : kHuge;
        u64 hide = ballot64(m4 < kHugeTest && m4 + qoff - E <= dk + E);
        while (hide) {
          const int64_t b = b0 + ctz64(hide);
          hide &= hide - 1;
          if (++scanned > 16) {
            gave_up = true;
            break;
          }
          const int64_t pos = (b >> 1) * 128 + 32 * (lane >> 4) + 8 * ((lane >> 2) & 3) + 4 * (b & 1) + (lane & 3);
          const bool valid = pos < w;
          const int r2 = valid ? ix.fi_sorted[grp.a + pos] : 0;
          L.cand_id[lane] = r2;
          WAVE_SYNC();
          const float d2 = wave_distances<METRIC>(ix, L.cand_id, L.cand_dist, L.qv, 64, 0);
          const u64 k2 = ((u64)fkey(d2) << 32) | ((u64)(uint32_t)r2 << 1);
          bool pass = valid;
          if (m >= K) pass = pass && ((k2 | 1ull) < (L.lbeam[K - 1] | 1ull));
          m = wave_merge(L.lbeam, m, K, pass, k2, L.cand_key, &p0);
        }
      }
      if (!gave_up) {
        for (int x = lane; x < m; x += 64) {
          const u64 e = L.lbeam[x];
          A.out_key[(size_t)ti * K + x] = (e & 0xffffffff00000000ull) | (uint32_t)((uint32_t)e >> 1);
        }
        outn = m;
        proven = true;
        if (lane == 0) atomicAdd(&ctr->gemm_rescued, 1ull);
      }
    }
    if (lane == 0) {
      A.out_cnt[ti] = outn;
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

int launch_gemm_scores(const GemmArgs &a, int num_cus, void *stream) {
  if (a.ix.stride > 128) {  // 129 .. 512 floats: slabs of 128, A operand in registers, one workgroup per CU
    if (a.ix.stride > 512 || (a.ix.stride & 15)) {
      g_gerr = "dimension too large for the dense prefilter tile";
      return 1;
    }
    const int slabs = (a.ix.stride + 127) / 128;
    // (four slabs: + the low halves of the last slab's A operand; the overlapped kernel: + its parity arrays and the raw half slab)
    const bool wide4 = slabs == 4;
    const size_t ldsw = (size_t)128 * (4 * 128 + 16) + (wide4 ? 4 : 3) * 128 * 4 + (slabs == 4 ? (size_t)4 * 8 * 64 * 16 : 0) + (wide4 ? (size_t)32 * 1024 : 0);
    void (*kw)(GemmArgs) = slabs == 2 ? k_gemm_scores_wide<2> : slabs == 3 ? k_gemm_scores_wide<3> : wide4 ? k_gemm_scores_wide4 : k_gemm_scores_wide<4>;
    if (gcheck(hipFuncSetAttribute((const void *)kw, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsw))) return 1;
    hipLaunchKernelGGL(kw, dim3(num_cus > 0 ? num_cus : 256), dim3(256), ldsw, (hipStream_t)stream, a);
    return gcheck(hipGetLastError());
  }
  const size_t lds = (size_t)128 * (4 * a.ix.stride + 16) + 3 * 128 * 4;
  void (*kern)(GemmArgs) = nullptr;
  switch (a.ix.stride) {
    case 16: kern = k_gemm_scores<16>; break;
    case 32: kern = k_gemm_scores<32>; break;
    case 48: kern = k_gemm_scores<48>; break;
    case 64: kern = k_gemm_scores<64>; break;
    case 80: kern = k_gemm_scores<80>; break;
    case 96: kern = k_gemm_scores<96>; break;
    case 112: kern = k_gemm_scores<112>; break;
    case 128: kern = k_gemm_scores<128>; break;
    default: g_gerr = "row stride is not a multiple of 16 floats"; return 1;
  }
  if (lds > 48 * 1024)
    if (gcheck(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds))) return 1;
  // two workgroups per CU (the LDS allows it): one stores its scores while the other runs its MFMAs
  hipLaunchKernelGGL(kern, dim3(2 * (num_cus > 0 ? num_cus : 256)), dim3(256), lds, (hipStream_t)stream, a);
  return gcheck(hipGetLastError());
}

int launch_select_rerank(const GemmArgs &a, Counters *ctr, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  const int blocks = (int)std::min<int64_t>(4096, (a.nq + 3) / 4);
  const size_t lds = (size_t)4 * (((a.ix.stride * 4 + 15) & ~15) + 64 * 8 + 64 * 4 + 64 * 4 + ((a.k + 1) & ~1) * 8);
  if (a.ix.metric == 1) hipLaunchKernelGGL(k_rerank<1>, dim3(blocks), dim3(256), lds, s, a, ctr);
  else hipLaunchKernelGGL(k_rerank<0>, dim3(blocks), dim3(256), lds, s, a, ctr);
  return gcheck(hipGetLastError());
}

}  // namespace wann
