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

// one workgroup (4 waves) per (group, 32-query tile); every wave owns a 32-point sub-tile of each
// 128-point chunk of the window
__global__ __launch_bounds__(256) void k_gemm_scores(GemmArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const IndexView &ix = A.ix;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int DP = ix.stride + 1;  // +1: rows land on different LDS banks
  float *Qs = reinterpret_cast<float *>(smem);        // [32][DP]
  float *qn = Qs + 32 * DP;                           // [32]
  float *Ps = qn + 32 + (size_t)wv * 32 * DP;         // [32][DP] per wave
  const GemmTile tile = A.tiles[blockIdx.x];
  const GemmGroup grp = A.groups[tile.group];
  const int64_t w = grp.b - grp.a;
  const int q0 = tile.q0;

  for (int idx = tid; idx < 32 * ix.stride; idx += 256) {
    const int r = idx / ix.stride, c = idx - r * ix.stride;
    float v = 0.f;
    if (q0 + r < grp.qcount && c < ix.d) v = A.queries[(int64_t)A.gq[grp.qoff + q0 + r] * ix.d + c];
    Qs[r * DP + c] = v;
  }
  __syncthreads();
  if (tid < 32) {
    float s = 0.f;
    for (int c = 0; c < ix.d; c++) s = fmaf(Qs[tid * DP + c], Qs[tid * DP + c], s);
    qn[tid] = s;
  }
  __syncthreads();

  const int half = lane >> 5, col = lane & 31;
  for (int64_t c0 = 0; c0 < w; c0 += 128) {
    const int64_t pbase = grp.a + c0 + 32 * wv;  // first window position of this wave's sub-tile
    for (int idx = lane; idx < 32 * ix.stride; idx += 64) {
      const int r = idx / ix.stride, c = idx - r * ix.stride;
      float v = 0.f;
      if (pbase + r < grp.b) v = ix.points[(int64_t)ix.fi_sorted[pbase + r] * ix.stride + c];
      Ps[r * DP + c] = v;
    }
    float pn = 0.f;
    if (pbase + col < grp.b) pn = A.pnorm2[ix.fi_sorted[pbase + col]];
    WAVE_SYNC();
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int kk = 0; kk < ix.stride; kk += 2) {  // padding columns are zero
      const float a = Qs[col * DP + kk + half];  // A[i = lane & 31][k = lane >> 5]
      const float b = Ps[col * DP + kk + half];  // B[k = lane >> 5][j = lane & 31]
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
#pragma unroll
    for (int reg = 0; reg < 16; reg++) {  // C/D: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
      const int row = (reg & 3) + 8 * (reg >> 2) + 4 * half;
      const int64_t pos = c0 + 32 * wv + col;
      if (q0 + row < grp.qcount && pos < w) {
        const float dot = acc[reg];
        const float score = (ix.metric == 1) ? -dot : (qn[row] + pn - 2.f * dot);
        A.scores[grp.soff + (int64_t)(q0 + row) * w + pos] = score;
      }
    }
    WAVE_SYNC();
  }
}

// one wave per grouped query: the kSelect best scores of its row of S
__global__ __launch_bounds__(256) void k_select_scores(GemmArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = lane_id(), wv = threadIdx.x >> 6;
  u64 *cand_key = reinterpret_cast<u64 *>(smem) + (size_t)wv * (64 + kSelect);
  u64 *top = cand_key + 64;
  for (int64_t tq = (int64_t)blockIdx.x * 4 + wv; tq < A.ntq; tq += (int64_t)gridDim.x * 4) {
    const GemmGroup grp = A.groups[A.tq_group[tq]];
    const int64_t w = grp.b - grp.a;
    const float *srow = A.scores + grp.soff + (int64_t)A.tq_local[tq] * w;
    int m = 0;
    for (int64_t c0 = 0; c0 < w; c0 += 64) {
      const int64_t pos = c0 + lane;
      const bool have = pos < w;
      const float sc = have ? srow[pos] : 0.f;
      const u64 key = ((u64)fkey(sc) << 32) | ((u64)(uint32_t)pos << 1);
      bool pass = have;
      if (pass && m >= kSelect) pass = (key | 1ull) < (top[kSelect - 1] | 1ull);
      int p0;
      m = wave_merge<u64 *, false>(top, m, kSelect, pass, key, cand_key, &p0);
    }
    if (lane < m) A.sel_pos[tq * kSelect + lane] = (int32_t)((uint32_t)top[lane] >> 1);
    if (lane == 0) {
      A.sel_cnt[tq] = m;
      A.sel_cut[tq] = (m == kSelect && w > kSelect) ? funkey((uint32_t)(top[kSelect - 1] >> 32)) : 3.402823466e+38f;
    }
    WAVE_SYNC();
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
  const int DP = a.ix.stride + 1;
  size_t lds = ((size_t)32 * DP + 32 + (size_t)4 * 32 * DP) * 4;
  auto kern = k_gemm_scores;
  if (lds > 48 * 1024)
    if (gcheck(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds))) return 1;
  hipLaunchKernelGGL(kern, dim3(a.ntiles), dim3(256), lds, (hipStream_t)stream, a);
  return gcheck(hipGetLastError());
}

int launch_select_rerank(const GemmArgs &a, void *stream) {
  if (a.ntq <= 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const int blocks = (int)std::min<int64_t>(4096, (a.ntq + 3) / 4);
  hipLaunchKernelGGL(k_select_scores, dim3(blocks), dim3(256), (size_t)4 * (64 + kSelect) * 8, s, a);
  if (gcheck(hipGetLastError())) return 1;
  const size_t lds = (size_t)4 * (((a.ix.stride * 4 + 15) & ~15) + 64 * 8 + 64 * 4 + 64 * 4);
  if (a.ix.metric == 1) hipLaunchKernelGGL(k_rerank<1>, dim3(blocks), dim3(256), lds, s, a);
  else hipLaunchKernelGGL(k_rerank<0>, dim3(blocks), dim3(256), lds, s, a);
  return gcheck(hipGetLastError());
}

}  // namespace wann
