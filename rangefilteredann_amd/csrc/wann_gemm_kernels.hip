// wann_gemm_kernels.hip -- the dense prefilter path: when many queries share one label window
// (PrefilterIndex::batch_search, src/prefiltering.h:124-204 -- e.g. the adversarial dataset, where 99
// queries share each 10 000-point window) the brute-force scan is a true Q x N contraction and runs
// on the matrix cores:
//
//   k_point_norms    |p|^2 of every point (once per index) and the largest of them
//   k_gemm_scores    per (window group, 32-query tile): S = Q . P^T with v_mfma_f32_32x32x2_f32
//                    (fp32 in / fp32 accumulate), scores -q.p (MIPS) or |q|^2 + |p|^2 - 2 q.p (L2)
//   k_select_scores  per query: the 32 best scores of its window
//   k_rerank         per query: exact reference-order distances of those 32 candidates, ordered by
//                    (dist, id), first k; plus a proof that no unselected point can belong to the
//                    top k (MFMA score error bound); queries that cannot be proven fall back to
//                    the exact scan kernel k_brute
//
// MFMA sums a k-ordered fmaf chain, the reference a different order, so MFMA scores
// only SELECT candidates; every returned distance is computed by the reference-order routines.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "wann_gemm_device.h"
#include "wann_wave.h"

namespace wann {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

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
    atomicMax(max_bits, __float_as_uint(s));  // s >= 0: the bit pattern orders like the value
  }
}

// One workgroup (4 waves, one per SIMD) per (group, 128-query tile, kGemmPointChunk-point slice of the
// window).  Per 128-point step the block stages a 128 x d point tile in the LDS next to the 128 x d query tile;
// every wave owns a 64 x 64 corner of the 128 x 128 score tile = 2 x 2 MFMA tiles (64 accumulator registers),
// so each LDS operand feeds two MFMAs.
template <int STRIDE>  // padded row length in floats: a multiple of 16, <= 128
__global__ __launch_bounds__(256) void k_gemm_scores(GemmArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const IndexView &ix = A.ix;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  constexpr int DP = STRIDE + 4;  // rows stay 16-byte aligned; 16 consecutive rows cover all 64 banks
  float *Qs = reinterpret_cast<float *>(smem);  // [128][DP]
  float *Ps = Qs + 128 * DP;                    // [128][DP]
  float *qn = Ps + 128 * DP;                    // [128]
  float *pn = qn + 128;                         // [128]
  const GemmTile tile = A.tiles[blockIdx.x];
  const GemmGroup grp = A.groups[tile.group];
  const int64_t w = grp.b - grp.a, wp = (w + 3) & ~(int64_t)3;  // score rows are padded to 16 bytes
  const int q0 = tile.q0;
  constexpr int s4 = STRIDE >> 2;

  int *rid = reinterpret_cast<int *>(pn + 128);  // [128] point rows of the tile being fetched
  constexpr int nit = s4 >> 1;                   // 128 rows x s4 float4 / 256 threads (s4 is even)

  // loads are unconditional (clamped indices, select afterwards): no dependent-load / branch chains
  const int dlast = ix.d - 1;
#pragma unroll 2
  for (int it = 0; it < nit; it++) {
    const int idx = tid + it * 256;
    const int r = idx / s4, c = (idx - r * s4) * 4;
    const int qr = (q0 + r < grp.qcount) ? (q0 + r) : (grp.qcount - 1);
    const float *src = A.queries + (int64_t)A.gq[grp.qoff + qr] * ix.d;
    float4 v;
    v.x = src[min(c + 0, dlast)]; v.y = src[min(c + 1, dlast)]; v.z = src[min(c + 2, dlast)]; v.w = src[min(c + 3, dlast)];
    const bool live = q0 + r < grp.qcount;
    v.x = (live && c + 0 < ix.d) ? v.x : 0.f;
    v.y = (live && c + 1 < ix.d) ? v.y : 0.f;
    v.z = (live && c + 2 < ix.d) ? v.z : 0.f;
    v.w = (live && c + 3 < ix.d) ? v.w : 0.f;
    *reinterpret_cast<float4 *>(Qs + r * DP + c) = v;
  }
  const int64_t wlast = w - 1;
  if (tid < 128) rid[tid] = ix.fi_sorted[grp.a + min(tile.p0 + tid, wlast)];
  __syncthreads();
  {
    const int r = tid >> 1, h = tid & 1;  // two threads per query row
    float sq = 0.f;
    for (int c = h; c < STRIDE; c += 2) sq = fmaf(Qs[r * DP + c], Qs[r * DP + c], sq);
    sq += __shfl_xor(sq, 1);
    if (h == 0) qn[r] = sq;
  }

  const int half = lane >> 5, col = lane & 31;
  const int wr = wv >> 1, wc = wv & 1;
  const int64_t pend = (tile.p0 + kGemmPointChunk < w) ? (tile.p0 + kGemmPointChunk) : w;
  // The next point tile travels HBM -> registers while the MFMA loop of the current one runs (one wave per
  // SIMD: the 512-register budget is all ours), and registers -> LDS after the barrier.  Its row numbers
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
  WANN_FETCH(tile.p0)
  for (int64_t c0 = tile.p0; c0 < pend; c0 += 128) {
    __syncthreads();  // previous step's reads of Ps / pn / rid are done (and qn is visible)
#pragma unroll
    for (int it = 0; it < nit; it++) {
      const int idx = tid + it * 256;
      const int r = idx / s4, c = (idx - r * s4) * 4;
      *reinterpret_cast<f32x4 *>(Ps + r * DP + c) = pre[it];
    }
    if (tid < 128) {
      pn[tid] = pre_n;
      rid[tid] = pre_rid;
    }
    __syncthreads();
    WANN_FETCH(c0 + 128)  // unconditional (row numbers are clamped): a conditional fetch would make the compiler wait for it here
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < 2; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    // The MFMA's two k slots are any two columns as long as A and B agree: lanes 0-31 take columns
    // kk .. kk+3 of an 8-column block and lanes 32-63 columns kk+4 .. kk+7, one ds_read_b128 per operand
    // for four MFMAs; the next block's operands are read while the current 16 MFMAs run.
    const float *qa = Qs + (64 * wr + col) * DP + 4 * half, *pb = Ps + (64 * wc + col) * DP + 4 * half;
    float4 a0n = *reinterpret_cast<const float4 *>(qa), a1n = *reinterpret_cast<const float4 *>(qa + 32 * DP);
    float4 b0n = *reinterpret_cast<const float4 *>(pb), b1n = *reinterpret_cast<const float4 *>(pb + 32 * DP);
#pragma unroll 2
    for (int kk = 0; kk < STRIDE; kk += 8) {  // stride is a multiple of 16; padding columns are zero
      const float4 a0 = a0n, a1 = a1n, b0 = b0n, b1 = b1n;
      if (kk + 8 < STRIDE) {
        a0n = *reinterpret_cast<const float4 *>(qa + kk + 8);
        a1n = *reinterpret_cast<const float4 *>(qa + 32 * DP + kk + 8);
        b0n = *reinterpret_cast<const float4 *>(pb + kk + 8);
        b1n = *reinterpret_cast<const float4 *>(pb + 32 * DP + kk + 8);
      }
#define WANN_MFMA4(C)                                                                   \
  acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.C, b0.C, acc[0][0], 0, 0, 0);     \
  acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.C, b1.C, acc[0][1], 0, 0, 0);     \
  acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.C, b0.C, acc[1][0], 0, 0, 0);     \
  acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.C, b1.C, acc[1][1], 0, 0, 0);
      WANN_MFMA4(x) WANN_MFMA4(y) WANN_MFMA4(z) WANN_MFMA4(w)
#undef WANN_MFMA4
    }
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < 2; j++)
#pragma unroll
        for (int reg = 0; reg < 16; reg++) {  // C/D: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
          const int row = 64 * wr + 32 * i + (reg & 3) + 8 * (reg >> 2) + 4 * half;
          const int pc = 64 * wc + 32 * j + col;
          const int64_t pos = c0 + pc;
          if (q0 + row < grp.qcount && pos < w) {
            const float dot = acc[i][j][reg];
            const float score = (ix.metric == 1) ? -dot : (qn[row] + pn[pc] - 2.f * dot);
            A.scores[grp.soff + (int64_t)(q0 + row) * wp + pos] = score;
          }
        }
  }
}

// one wave per grouped query: the kSelect best scores of its row of S
#undef WANN_FETCH


// One wave per grouped query.  The kSelect best (score, position) keys live sorted in the registers of
// lanes 0 .. kSelect-1; a row is streamed 1024 scores at a time (four 16-byte loads per lane in flight)
// and the few scores below the current cut are inserted one by one with a ballot + one-lane shift.
__global__ __launch_bounds__(256) void k_select_scores(GemmArgs A) {
  const int lane = lane_id(), wv = threadIdx.x >> 6;
  for (int64_t tq = (int64_t)blockIdx.x * 4 + wv; tq < A.ntq; tq += (int64_t)gridDim.x * 4) {
    const GemmGroup grp = A.groups[A.tq_group[tq]];
    const int64_t w = grp.b - grp.a, wp = (w + 3) & ~(int64_t)3;
    const float *srow = A.scores + grp.soff + (int64_t)A.tq_local[tq] * wp;
    u64 top = ~0ull, thr = ~0ull;  // ~0 = empty slot; thr = key in lane kSelect-1
    for (int64_t c0 = 0; c0 < wp; c0 += 1024) {
      f32x4 v[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int64_t off = c0 + j * 256 + 4 * lane;
        v[j] = (off < wp) ? *reinterpret_cast<const f32x4 *>(srow + off) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int cmp = 0; cmp < 4; cmp++) {
          const int64_t pos = c0 + j * 256 + 4 * lane + cmp;
          const u64 key = ((u64)fkey(v[j][cmp]) << 32) | ((u64)(uint32_t)pos << 1);
          u64 mask = ballot64(pos < w && key < thr);
          while (mask) {
            const u64 ck = rdlane64(key, ctz64(mask));
            mask &= mask - 1;
            if (ck < thr) {  // wave-uniform; thr may have dropped since the ballot
              const int p = popc64(ballot64(top < ck));  // top is sorted: a prefix of the lanes
              const u64 up = wave_shr1(top);
              if (lane < kSelect) top = (lane == p) ? ck : (lane > p ? up : top);
              thr = rdlane64(top, kSelect - 1);
            }
          }
        }
    }
    const int m = popc64(ballot64(top != ~0ull));
    if (lane < m) A.sel_pos[tq * kSelect + lane] = (int32_t)((uint32_t)top >> 1);
    if (lane == 0) {
      A.sel_cnt[tq] = m;
      A.sel_cut[tq] = (m == kSelect && w > kSelect) ? funkey((uint32_t)(thr >> 32)) : 3.402823466e+38f;
    }
  }
}

// one wave per grouped query: exact distances of the selected candidates, (dist, id) order, proof
template <int METRIC>
__global__ __launch_bounds__(256) void k_rerank(GemmArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const IndexView &ix = A.ix;
  const int lane = lane_id(), wv = threadIdx.x >> 6;
  const int per_wave = wave_lds_common_bytes(ix.stride);
  const WaveLds L = carve_wave_lds(smem + (size_t)wv * per_wave, ix.stride, 0, true);
  for (int64_t tq = (int64_t)blockIdx.x * 4 + wv; tq < A.ntq; tq += (int64_t)gridDim.x * 4) {
    const GemmGroup grp = A.groups[A.tq_group[tq]];
    const int qrow = A.gq[grp.qoff + A.tq_local[tq]];
    const int cnt = A.sel_cnt[tq];
    float q2 = 0.f;
    for (int i = lane; i < ix.stride; i += 64) {
      const float v = (i < ix.d) ? A.queries[(int64_t)qrow * ix.d + i] : 0.f;
      L.qv[i] = v;
      q2 = fmaf(v, v, q2);
    }
    for (int o = 32; o > 0; o >>= 1) q2 += __shfl_xor(q2, o);
    int rid = 0;
    if (lane < cnt) rid = ix.fi_sorted[grp.a + A.sel_pos[tq * kSelect + lane]];
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
    // proof: every unselected point has score >= cut, and |score - exact| <= E
    const float pmax = __uint_as_float(*A.pnorm2_max_bits);
    const float eps = 8.f * (float)(ix.d + 8) * 5.9604645e-8f;
    const float E = (METRIC == 1) ? eps * sqrtf(q2 * pmax) : 2.f * eps * (q2 + pmax);
    const int kk = cnt < A.k ? cnt : A.k;
    float dk = -3.402823466e+38f;  // k-th exact distance (the worst one that is returned)
    {
      const u64 hit = ballot64(lane < cnt && rank == kk - 1);
      if (hit) dk = __shfl(dist, ctz64(hit));
    }
    const float cut = A.sel_cut[tq];
    const bool proven = (cut == 3.402823466e+38f) || (dk + E < cut - E);
    if (lane == 0) {
      A.out_cnt[ti] = kk;
      if (!proven) A.fallback_list[atomicAdd(A.fallback_count, 1)] = ti;
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

int launch_gemm_scores(const GemmArgs &a, void *stream) {
  if (a.ntiles <= 0) return 0;
  const int DP = a.ix.stride + 4;
  size_t lds = ((size_t)2 * 128 * DP + 384) * 4;
  if (lds > 160 * 1024 || a.ix.stride > 128) {
    g_gerr = "dimension too large for the dense prefilter tile";
    return 1;
  }
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
  hipLaunchKernelGGL(kern, dim3(a.ntiles), dim3(256), lds, (hipStream_t)stream, a);
  return gcheck(hipGetLastError());
}

int launch_select_rerank(const GemmArgs &a, void *stream) {
  if (a.ntq <= 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const int blocks = (int)std::min<int64_t>(4096, (a.ntq + 3) / 4);
  hipLaunchKernelGGL(k_select_scores, dim3(blocks), dim3(256), 0, s, a);
  if (gcheck(hipGetLastError())) return 1;
  const size_t lds = (size_t)4 * (((a.ix.stride * 4 + 15) & ~15) + 64 * 8 + 64 * 4 + 64 * 4);
  if (a.ix.metric == 1) hipLaunchKernelGGL(k_rerank<1>, dim3(blocks), dim3(256), lds, s, a);
  else hipLaunchKernelGGL(k_rerank<0>, dim3(blocks), dim3(256), lds, s, a);
  return gcheck(hipGetLastError());
}

}  // namespace wann
