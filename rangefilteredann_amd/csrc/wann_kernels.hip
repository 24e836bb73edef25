// wann_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the window-filtered ANN engine.
//
// One wavefront owns one unit of work (a beam search of one query in one partition, or a
// brute-force scan); a 256-thread workgroup holds four independent waves that never synchronise
// with each other.  All cross-lane traffic is ballots / readlane / per-wave LDS.  Kernels are
// persistent over a device-side work list (atomic cursor), so no launch depends on a host count.
//
//   k_route     window -> work item (window search tree descent)          reference: src/range_filter_tree.h:403-471,
//                                                                          src/super_optimized_postfilter_tree.h:187-270
//   k_search    batched beam search + post filter of the final beam       reference: ParlayANN/algorithms/utils/beamSearch.h:51-184,
//                                                                          src/postfilter_vamana.h:141-188,223-254
//   k_brute     exact scan of a contiguous (or gathered) window            reference: src/range_filter_tree.h:393-397, src/prefiltering.h:154-204
//   k_finalize  top-k rows -> (ids, dists) with id decoding and padding    reference: src/range_filter_tree.h:84-93
//
// fp32 evaluation order is the reference's as compiled; this file is built with
// -ffp-contract=off and every fused multiply-add below is an explicit fmaf().
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "wann_device.h"

namespace wann {

typedef unsigned long long u64;

#define WAVE_SYNC()                                            \
  do {                                                         \
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");     \
    __builtin_amdgcn_wave_barrier();                           \
  } while (0)

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ u64 ballot64(bool p) { return __ballot(p); }
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int rdlane(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ u64 rdlane64(u64 v, int l) {
  uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
  uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l);
  return ((u64)hi << 32) | lo;
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

// Negative inner product (mips_point.h:60-66 as compiled): running scalar, products rounded then
// added in index order for the first 8*floor(d/8) elements, fused for the tail.  One lane per row.
template <int NB>
__device__ __forceinline__ float mips_lane(const float *__restrict__ prow, const float *qv, int d,
                                           bool active) {
  float r = 0.f;
  if (active) {
    const int nch = (d + 3) >> 2;
    const int dv = d & ~7;
    for (int c0 = 0; c0 < nch; c0 += NB) {
      float4 buf[NB];
#pragma unroll
      for (int j = 0; j < NB; j++) {
        int c = c0 + j;
        if (c < nch) buf[j] = *reinterpret_cast<const float4 *>(prow + 4 * c);
      }
#pragma unroll
      for (int j = 0; j < NB; j++) {
        int c = c0 + j;
        if (c < nch) {
          float4 q = *reinterpret_cast<const float4 *>(qv + 4 * c);
          if (4 * c + 3 < dv) {
            r = __fadd_rn(r, __fmul_rn(q.x, buf[j].x));
            r = __fadd_rn(r, __fmul_rn(q.y, buf[j].y));
            r = __fadd_rn(r, __fmul_rn(q.z, buf[j].z));
            r = __fadd_rn(r, __fmul_rn(q.w, buf[j].w));
          } else {  // dv is a multiple of 8, so a chunk is entirely vector part or entirely tail
            r = fmaf(q.x, buf[j].x, r);
            r = fmaf(q.y, buf[j].y, r);
            r = fmaf(q.z, buf[j].z, r);
            r = fmaf(q.w, buf[j].w, r);
          }
        }
      }
    }
  }
  return -r;
}

// Distances of `cnt` rows whose (sorted-order) row numbers sit in ids_lds[0..cnt): afterwards lane
// s < cnt holds the distance of row s.  scratch_lds: 64 floats.
template <int METRIC>
__device__ __forceinline__ float wave_distances(const IndexView &ix, const int32_t *ids_lds,
                                                float *scratch_lds, const float *qv, int cnt,
                                                int64_t row_off) {
  const int lane = lane_id();
  if (METRIC == 1) {
    bool act = lane < cnt;
    int id = act ? ids_lds[lane] : 0;
    const float *prow = ix.points + (row_off + id) * (int64_t)ix.stride;
    return mips_lane<8>(prow, qv, ix.d, act);
  } else {
    const int D8 = (ix.d + 7) >> 3;
    const int h = lane & 1;
    for (int base = 0; base < cnt; base += 32) {
      int s = base + (lane >> 1);
      bool act = s < cnt;
      int id = act ? ids_lds[s] : 0;
      const float *prow = ix.points + (row_off + id) * (int64_t)ix.stride;
      float dist = l2_pair<8>(prow, qv, D8, h, act);
      if (act && h) scratch_lds[s] = dist;
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
template <typename BeamPtr>
__device__ __forceinline__ int wave_merge(BeamPtr beam, int m, int B, bool pass, u64 key,
                                          u64 *cand_key, int *first_pos) {
  const int lane = lane_id();
  *first_pos = m;
  u64 smask = ballot64(pass);
  if (smask == 0) return m;
  int rank = 0;
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
  const int pos = lo;
  // std::set_union keeps max(copies in beam, copies among candidates) of equal elements: the j-th
  // copy of a candidate key is dropped iff the beam already holds more than j copies.  (Copies
  // arise when a row lists a node twice -- the reference's builder can append the start point
  // twice -- and the lossy filter lets both through.)
  bool dup = false;
  if (mine) {
    int j = 0, bx = 0;
    for (int l = lane - 1; l >= 0 && cand_key[l] == ck; l--) j++;
    while (pos + bx < m && ((beam[pos + bx] | 1ull) == (ck | 1ull))) bx++;
    dup = j < bx;
  }
  WAVE_SYNC();
  const u64 nd = ballot64(mine && !dup);
  const int cp = popc64(nd);
  if (cp == 0) return m;
  const int pre = popc64(nd & lanemask_lt());
  const int p0 = rdlane(pos, ctz64(nd));
  const int span = m - p0;
  if (span > 0) {
    for (int base = p0 + ((span - 1) & ~63); base >= p0; base -= 64) {
      int x = base + lane;
      bool act = x < m;
      u64 e = act ? beam[x] : 0ull;
      int sx = 0;
      for (u64 mm = nd; mm; mm &= mm - 1) {
        int pl = rdlane(pos, ctz64(mm));
        sx += (pl <= x) ? 1 : 0;
      }
      WAVE_SYNC();
      int nx = x + sx;
      if (act && nx < B) beam[nx] = e;
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
// k_search
// --------------------------------------------------------------------------------------------
__device__ __forceinline__ int lds_bytes_per_wave(int B, int bits, int stride, bool lds_table,
                                                  bool lds_beam) {
  int bytes = stride * 4;              // query vector
  bytes = (bytes + 15) & ~15;
  bytes += 64 * 8;                     // cand_key
  bytes += 64 * 4;                     // cand_id
  bytes += 64 * 4;                     // cand_dist
  if (lds_beam) bytes += ((B + 1) & ~1) * 8;
  if (lds_table) bytes += 4 << bits;
  return (bytes + 15) & ~15;
}

template <int METRIC, bool TABLE_LDS, bool BEAM_LDS>
__global__ __launch_bounds__(64 * kWavesPerBlock) void k_search(SearchArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const IndexView &ix = A.ix;
  const int lane = lane_id();
  const int wib = threadIdx.x >> 6;
  const int slot = blockIdx.x * kWavesPerBlock + wib;
  const int B = A.B;
  const int bits = A.bits;
  const int per_wave = lds_bytes_per_wave(B, bits, ix.stride, TABLE_LDS, BEAM_LDS);
  unsigned char *base = smem + (size_t)wib * per_wave;
  float *qv = reinterpret_cast<float *>(base);
  int off = (ix.stride * 4 + 15) & ~15;
  u64 *cand_key = reinterpret_cast<u64 *>(base + off);
  off += 64 * 8;
  int32_t *cand_id = reinterpret_cast<int32_t *>(base + off);
  off += 64 * 4;
  float *cand_dist = reinterpret_cast<float *>(base + off);
  off += 64 * 4;
  u64 *lbeam = reinterpret_cast<u64 *>(base + off);
  if (BEAM_LDS) off += ((B + 1) & ~1) * 8;
  int32_t *ltable = reinterpret_cast<int32_t *>(base + off);

  const uint32_t tmask = (1u << bits) - 1u;
  const int total = *A.list_count;
#define TRACE(v)                                                                                   \
  do {                                                                                             \
    if (A.trace && lane == 0)                                                                      \
      __hip_atomic_store(&A.trace[slot], (unsigned int)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); \
  } while (0)
  TRACE(1);

  for (;;) {
    const int t = wave_ticket(A.cursor);
    TRACE(0x100 + t);
    if (t >= total) break;
    const int ti = A.list[t];
    const Task task = A.tasks[ti];
    const PartDesc part = ix.parts[task.part];
    TRACE(2);
    const int64_t qrow = task.query;
    const int64_t qid = A.raw ? A.raw_qids[qrow] : (A.qid_base + qrow);
    const int64_t row_off = part.start;

    // stage the query (zero padded) and reset the seen-filter
    for (int i = lane; i < ix.stride; i += 64) qv[i] = (i < ix.d) ? A.queries[qrow * ix.d + i] : 0.f;
    if (TABLE_LDS) {
      for (int i = lane; i < (1 << bits); i += 64) ltable[i] = -1;
    } else {
      int4 *gt = reinterpret_cast<int4 *>(A.g_table + ((size_t)slot << bits));
      for (int i = lane; i < (1 << (bits - 2)); i += 64) gt[i] = make_int4(-1, -1, -1, -1);
    }
    WAVE_SYNC();

    TRACE(3);
    // frontier = {start node 0} (beamSearch.h:80-82)
    if (lane == 0) cand_id[0] = 0;
    WAVE_SYNC();
    float d0 = wave_distances<METRIC>(ix, cand_id, cand_dist, qv, 1, row_off);
    d0 = __shfl(d0, 0);
    TRACE(4);
    int m = 1, p = 0;
    long long nvis = 0, ncmp = 1;

    auto beam_ld = [&](int i) -> u64 {
      if (BEAM_LDS) return lbeam[i];
      return A.g_beam[(size_t)slot * A.g_beam_cap + i];
    };
    auto beam_st = [&](int i, u64 v) {
      if (BEAM_LDS) lbeam[i] = v;
      else A.g_beam[(size_t)slot * A.g_beam_cap + i] = v;
    };
    if (lane == 0) beam_st(0, ((u64)fkey(d0) << 32));
    WAVE_SYNC();

    while (p < m && nvis < A.limit) {
      TRACE(0x10000 + (int)nvis);
      // ---- visit the closest unvisited beam entry (beamSearch.h:111-117)
      const u64 curkey = beam_ld(p);
      const int cur = (int)((uint32_t)curkey >> 1);
      if (lane == 0) beam_st(p, curkey | 1ull);
      nvis++;

      // ---- adjacency row, coalesced (graph.h:198); -1 = unused slot
      int a = -1;
      if (lane < ix.rs) a = ix.graph[(part.row_base + cur) * (int64_t)ix.rs + lane];
      bool valid = (a >= 0) && (lane < A.degree_limit) && ((int64_t)a != qid);

      // ---- lossy direct-mapped "seen" filter, sequential semantics emulated exactly
      //      (beamSearch.h:68-73,126-131): lane i sees the id left in its slot by the nearest
      //      preceding lane of the row that hashed to the same slot, else the table's old value;
      //      the last lane of each slot class leaves its id in the table.
      const uint32_t loc = (uint32_t)hash64_2((u64)(uint32_t)a) & tmask;
      int old = -1;
      if (valid) old = TABLE_LDS ? ltable[loc] : A.g_table[((size_t)slot << bits) + loc];
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
      const bool seen = valid && (prev_val == a);
      WAVE_SYNC();
      if (valid && higher == 0) {
        if (TABLE_LDS) ltable[loc] = a;
        else A.g_table[((size_t)slot << bits) + loc] = a;
      }
      const bool keep = valid && !seen;
      const u64 kmask = ballot64(keep);
      const int nk = popc64(kmask);
      if (keep) cand_id[popc64(kmask & lanemask_lt())] = a;
      WAVE_SYNC();
      ncmp += nk;

      // ---- score the kept neighbours (beamSearch.h:135-145)
      float cutoff = 2147483648.0f;  // (float)INT_MAX
      if (m >= B) cutoff = funkey((uint32_t)(beam_ld(m - 1) >> 32));
      float dist = wave_distances<METRIC>(ix, cand_id, cand_dist, qv, nk, row_off);
      int cid = (lane < nk) ? cand_id[lane] : 0;
      WAVE_SYNC();
      const bool pass = (lane < nk) && (dist < cutoff);
      const u64 key = ((u64)fkey(dist) << 32) | ((u64)(uint32_t)cid << 1);

      // ---- sort + set_union + truncate (beamSearch.h:148-157)
      int p0;
      if (BEAM_LDS) m = wave_merge(lbeam, m, B, pass, key, cand_key, &p0);
      else m = wave_merge(A.g_beam + (size_t)slot * A.g_beam_cap, m, B, pass, key, cand_key, &p0);

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
    }

    TRACE(5);
    if (lane == 0) {
      atomicAdd(&A.ctr->beam_searches, 1ull);
      atomicAdd(&A.ctr->hops, (unsigned long long)nvis);
      atomicAdd(&A.ctr->dist_cmps, (unsigned long long)ncmp);
    }

    if (A.raw) {  // dump the whole beam (ids local to the partition)
      for (int x = lane; x < m; x += 64) {
        u64 e = beam_ld(x);
        A.raw_ids[qrow * B + x] = (int)((uint32_t)e >> 1);
        A.raw_dists[qrow * B + x] = funkey((uint32_t)(e >> 32));
      }
      if (lane == 0) {
        A.raw_sizes[qrow] = m;
        A.raw_hops[qrow] = nvis;
        A.raw_cmps[qrow] = ncmp;
      }
      TRACE(6);
      continue;
    }

    // ---- post filter: keep beam entries whose label lies in [lo,hi], first k of them
    //      (postfilter_vamana.h:234-251); ids become sorted-order indices (subset[local])
    int found = 0;
    long long labs = 0;
    for (int bx = 0; bx < m && found < A.k; bx += 64) {
      int x = bx + lane;
      bool act = x < m;
      u64 e = act ? beam_ld(x) : 0ull;
      int lid = (int)((uint32_t)e >> 1);
      float lab = act ? ix.labels[row_off + lid] : 0.f;
      bool inw = act && (lab >= task.lo) && (lab <= task.hi);
      u64 im = ballot64(inw);
      int idx = found + popc64(im & lanemask_lt());
      if (inw && idx < A.k)
        A.out_key[(size_t)ti * A.k + idx] = (e & 0xffffffff00000000ull) | (uint32_t)(row_off + lid);
      found += popc64(im);
      labs += (m - bx) < 64 ? (m - bx) : 64;
    }
    if (found > A.k) found = A.k;
    if (lane == 0) {
      A.out_cnt[ti] = found;
      atomicAdd(&A.ctr->label_reads, (unsigned long long)labs);
      if (!A.is_final) {
        if (found >= A.k) {  // doubling loop ends here (postfilter_vamana.h:161-172)
          if (A.wants_final) A.final_list[atomicAdd(A.final_count, 1)] = ti;
        } else if (A.can_double) {
          A.next_list[atomicAdd(A.next_count, 1)] = ti;
        }
      }
    }
  }
}

// --------------------------------------------------------------------------------------------
// k_brute: exact top-k over rows [a,b) of the sorted order (T_BRUTE) or over the label-argsort
// positions [a,b) of an unsorted point set (T_BRUTE_GATHER, prefiltering.h:189-194)
// --------------------------------------------------------------------------------------------
template <int METRIC>
__global__ __launch_bounds__(64 * kWavesPerBlock) void k_brute(BruteArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const IndexView &ix = A.ix;
  const int lane = lane_id();
  const int wib = threadIdx.x >> 6;
  const int K = A.k;
  int per_wave = ((ix.stride * 4 + 15) & ~15) + 64 * 8 + 64 * 4 + 64 * 4 + ((K + 1) & ~1) * 8;
  per_wave = (per_wave + 15) & ~15;
  unsigned char *base = smem + (size_t)wib * per_wave;
  float *qv = reinterpret_cast<float *>(base);
  int off = (ix.stride * 4 + 15) & ~15;
  u64 *cand_key = reinterpret_cast<u64 *>(base + off);
  off += 64 * 8;
  int32_t *cand_id = reinterpret_cast<int32_t *>(base + off);
  off += 64 * 4;
  float *cand_dist = reinterpret_cast<float *>(base + off);
  off += 64 * 4;
  u64 *top = reinterpret_cast<u64 *>(base + off);
  const int total = *A.list_count;
  const int step = (METRIC == 1) ? 64 : 64;

  for (;;) {
    const int t = wave_ticket(A.cursor);
    if (t >= total) break;
    const int ti = A.list[t];
    const Task task = A.tasks[ti];
    const int64_t qrow = task.query;
    for (int i = lane; i < ix.stride; i += 64) qv[i] = (i < ix.d) ? A.queries[qrow * ix.d + i] : 0.f;
    WAVE_SYNC();
    int m = 0;
    for (int64_t r0 = task.a; r0 < task.b; r0 += step) {
      int cnt = (int)((task.b - r0) < step ? (task.b - r0) : step);
      int64_t row = r0 + lane;
      int rid = 0;
      if (lane < cnt) rid = (task.mode == T_BRUTE_GATHER) ? ix.fi_sorted[row] : (int)row;
      cand_id[lane] = rid;
      WAVE_SYNC();
      float dist = wave_distances<METRIC>(ix, cand_id, cand_dist, qv, cnt, 0);
      u64 key = ((u64)fkey(dist) << 32) | ((u64)(uint32_t)rid << 1);
      bool pass = lane < cnt;
      if (m >= K) pass = pass && ((key | 1ull) < (top[K - 1] | 1ull));
      int p0;
      m = wave_merge(top, m, K, pass, key, cand_key, &p0);
    }
    for (int x = lane; x < m; x += 64) {
      u64 e = top[x];
      A.out_key[(size_t)ti * K + x] = (e & 0xffffffff00000000ull) | (uint32_t)((uint32_t)e >> 1);
    }
    if (lane == 0) {
      A.out_cnt[ti] = m;
      atomicAdd(&A.ctr->brute_rows, (unsigned long long)(task.b > task.a ? task.b - task.a : 0));
    }
    WAVE_SYNC();
  }
}

// --------------------------------------------------------------------------------------------
// k_route: one thread per query
// --------------------------------------------------------------------------------------------
__device__ __forceinline__ int64_t first_ge(const float *fv, int64_t n, float v) {  // tree_utils.h:19-37
  if (fv[0] >= v) return 0;
  int64_t s = 0, e = n;
  while (s + 1 < e) {
    int64_t mid = (s + e) / 2;
    if (fv[mid] >= v) e = mid;
    else s = mid;
  }
  return e;
}

// prefiltering.h:159-184 (r = n-1: the last point can never be selected)
__device__ __forceinline__ int64_t prefilter_bound(const float *fv, int64_t n, float v) {
  int64_t l = 0, r = n - 1;
  while (l < r) {
    int64_t mid = (l + r) / 2;
    if (fv[mid] < v) l = mid + 1;
    else r = mid;
  }
  return l;
}

// true when find_largest_ranges_within_query_range (range_filter_tree.h:234-295) yields a centre
__device__ bool has_centre(const IndexView &ix, uint64_t istart, uint64_t eend) {
  const uint64_t range_size = eend - istart;
  int row = -1;
  for (int r = 0; r < ix.nlevels; r++) {
    const int64_t *off = ix.wst_off + ix.wst_ptr[r];
    uint64_t bsz = (uint64_t)(off[1] - off[0] - 1);
    if (bsz <= range_size) {
      row = r;
      break;
    }
  }
  if (row < 0) return false;
  for (int attempt = 0; attempt < 2; attempt++) {
    const int64_t *off = ix.wst_off + ix.wst_ptr[row];
    const int64_t nb = ix.level_nb[row];
    int64_t first = 0;
    if (istart != 0) {  // bucket containing istart-1, plus one
      int64_t lo = 0, hi = nb;  // largest b with off[b] <= istart-1
      while (lo + 1 < hi) {
        int64_t mid = (lo + hi) / 2;
        if ((uint64_t)off[mid] <= istart - 1) lo = mid;
        else hi = mid;
      }
      first = lo + 1;
    }
    if (first >= nb) return attempt == 0 ? false : false;  // reference would index past the row
    uint64_t end = (uint64_t)off[first + 1];
    if (end <= eend) return true;
    if (attempt == 1) return true;  // second row is taken as is (range_filter_tree.h:268-281)
    row += 1;
    if (row >= ix.nlevels) return false;
  }
  return true;
}

__global__ void k_route(RouteArgs A) {
  const IndexView &ix = A.ix;
  int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= A.nq) return;
  const float lo = A.ranges[2 * q], hi = A.ranges[2 * q + 1];
  Task t;
  t.query = (int32_t)q;
  t.mode = T_EMPTY;
  t.part = 0;
  t.flags = 0;
  t.a = t.b = 0;
  t.lo = lo;
  t.hi = hi;
  const bool beam_ok = A.beam < A.max_beam;  // postfilter_vamana.h:161: no search at all otherwise

  if (ix.kind == 0) {  // PrefilterIndex: [lb(lo), lb(hi)) over the label argsort
    t.a = prefilter_bound(ix.fv_sorted, ix.n, lo);
    t.b = prefilter_bound(ix.fv_sorted, ix.n, hi);
    t.mode = (t.b > t.a) ? T_BRUTE_GATHER : T_EMPTY;
  } else if (ix.kind == 1) {  // stand-alone PostfilterVamanaIndex: always the one graph
    t.mode = beam_ok ? T_GRAPH : T_EMPTY;
    t.part = 0;
  } else {
    const bool empty = hi < ix.labels[0] || lo > ix.labels[ix.n - 1];  // range_filter_tree.h:191-203
    if (!empty) {
      const uint64_t istart = (uint64_t)first_ge(ix.labels, ix.n, lo);
      const uint64_t eend = (uint64_t)first_ge(ix.labels, ix.n, hi);
      const uint64_t w = eend - istart;
      int level = 0;
      int64_t idx = 0;
      bool brute = false, general = false;
      if (ix.kind == 4) {  // super tree (super_optimized_postfilter_tree.h:204-243)
        for (level = ix.nlevels - 1; level >= 0; level--) {
          if (level == 0) {
            idx = 0;
            break;
          }
          uint64_t bsz = (uint64_t)ix.sup_size[level];
          if (bsz < w) continue;
          uint64_t shift = (uint64_t)ix.sup_shift[level];
          uint64_t nb = (uint64_t)ix.level_nb[level];
          uint64_t fp = istart / shift, lp = (eend - 1) / shift;
          if (fp > nb - 1) fp = nb - 1;
          if (lp > nb - 1) lp = nb - 1;
          bool found = false;
          for (uint64_t tb = fp; tb <= lp; tb++) {
            uint64_t bs = tb * shift, be = bs + bsz;
            if (be > (uint64_t)ix.n) be = (uint64_t)ix.n;
            if (istart >= bs && eend <= be) {
              idx = (int64_t)tb;
              found = true;
              break;
            }
          }
          if (found) break;
        }
      } else if (A.method != M_OPTIMIZED) {
        general = true;  // fenwick / three_split: multi-bucket cover
      } else {
        if (4 * w < (uint64_t)(int64_t)ix.cutoff) {  // range_filter_tree.h:419-421 -> fenwick
          if (has_centre(ix, istart, eend)) general = true;
          else brute = true;
        } else {
          int64_t row = 0;
          idx = 0;
          while (row + 1 < ix.nlevels) {  // :426-451
            const int64_t nrow = row + 1;
            const int64_t *off = ix.wst_off + ix.wst_ptr[nrow];
            int64_t nidx = -1;
            for (int64_t c = idx * ix.split; c < idx * ix.split + ix.split; c++) {
              if (c >= ix.level_nb[nrow]) break;
              if (istart >= (uint64_t)off[c] && eend <= (uint64_t)off[c + 1]) nidx = c;
            }
            if (nidx < 0) break;
            idx = nidx;
            row = nrow;
          }
          level = (int)row;
          if (A.has_ratio) {  // :460-466
            const int64_t *off = ix.wst_off + ix.wst_ptr[row];
            float ratio = (float)(uint64_t)(off[idx + 1] - off[idx]) / (float)w;
            if (ratio > A.ratio) {
              if (has_centre(ix, istart, eend)) general = true;
              else brute = true;
            }
          }
        }
      }
      if (general) {
        atomicAdd(&A.ctr->unsupported, 1ull);
      } else if (brute) {
        t.a = (int64_t)istart;
        t.b = (int64_t)eend;
        t.mode = (eend > istart) ? T_BRUTE : T_EMPTY;
      } else {
        const int32_t pidx = (int32_t)(ix.level_part0[level] + idx);
        if (ix.vamana_leaves) {
          t.mode = beam_ok ? T_GRAPH : T_EMPTY;
          t.part = pidx;
        } else {  // PrefilterIndex leaf on a slice of the sorted order
          const PartDesc pd = ix.parts[pidx];
          int64_t s = prefilter_bound(ix.labels + pd.start, pd.n, lo);
          int64_t e = prefilter_bound(ix.labels + pd.start, pd.n, hi);
          t.a = pd.start + s;
          t.b = pd.start + e;
          t.mode = (e > s) ? T_BRUTE : T_EMPTY;
        }
      }
    }
  }
  A.tasks[q] = t;
  if (t.mode == T_GRAPH) A.graph_list[atomicAdd(A.graph_count, 1)] = (int32_t)q;
  else if (t.mode == T_BRUTE || t.mode == T_BRUTE_GATHER) A.brute_list[atomicAdd(A.brute_count, 1)] = (int32_t)q;
}

__global__ void k_finalize(FinalizeArgs A) {
  int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= A.nq) return;
  const Task t = A.tasks[q];
  int cnt = (t.mode == T_EMPTY) ? 0 : A.out_cnt[q];
  const bool decode = A.decode && t.mode != T_BRUTE_GATHER;
  for (int j = 0; j < A.k; j++) {
    uint32_t id = A.pad_id;
    float dist = 3.402823466e+38f;  // std::numeric_limits<float>::max()
    if (j < cnt) {
      u64 e = A.out_key[(size_t)q * A.k + j];
      id = (uint32_t)e;
      if (decode) id = A.ix.decoding[id];
      dist = funkey((uint32_t)(e >> 32));
    }
    A.ids[q * A.k + j] = id;
    A.dists[q * A.k + j] = dist;
  }
}

// --------------------------------------------------------------------------------------------
// launchers
// --------------------------------------------------------------------------------------------
static thread_local const char *g_launch_err = "";
const char *launch_last_error() { return g_launch_err; }

static int check(hipError_t e) {
  if (e != hipSuccess) {
    g_launch_err = hipGetErrorString(e);
    return 1;
  }
  return 0;
}

int search_lds_bytes_per_wave(int B, int bits, int stride, int lds_table, int lds_beam) {
  int bytes = (stride * 4 + 15) & ~15;
  bytes += 64 * 8 + 64 * 4 + 64 * 4;
  if (lds_beam) bytes += ((B + 1) & ~1) * 8;
  if (lds_table) bytes += 4 << bits;
  return (bytes + 15) & ~15;
}

int launch_route(const RouteArgs &a, void *stream) {
  if (a.nq == 0) return 0;
  int threads = 128;
  int blocks = (int)((a.nq + threads - 1) / threads);
  hipLaunchKernelGGL(k_route, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, a);
  return check(hipGetLastError());
}

template <int METRIC>
static int launch_search_m(const SearchArgs &a, const LaunchCfg &cfg, void *stream) {
  size_t lds = (size_t)search_lds_bytes_per_wave(a.B, a.bits, a.ix.stride, cfg.lds_table, cfg.lds_beam) * kWavesPerBlock;
  dim3 grid(cfg.blocks), block(64 * kWavesPerBlock);
  hipStream_t s = (hipStream_t)stream;
#define WANN_LAUNCH(TL, BL)                                                                     \
  do {                                                                                          \
    auto kern = k_search<METRIC, TL, BL>;                                                       \
    if (lds > 48 * 1024)                                                                        \
      if (check(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds))) return 1; \
    hipLaunchKernelGGL(kern, grid, block, lds, s, a);                                           \
  } while (0)
  if (cfg.lds_table && cfg.lds_beam) WANN_LAUNCH(true, true);
  else if (!cfg.lds_table && cfg.lds_beam) WANN_LAUNCH(false, true);
  else if (cfg.lds_table && !cfg.lds_beam) WANN_LAUNCH(true, false);
  else WANN_LAUNCH(false, false);
#undef WANN_LAUNCH
  return check(hipGetLastError());
}

int launch_search(const SearchArgs &a, const LaunchCfg &cfg, void *stream) {
  if (cfg.blocks <= 0) return 0;
  return a.ix.metric == 1 ? launch_search_m<1>(a, cfg, stream) : launch_search_m<0>(a, cfg, stream);
}

int launch_brute(const BruteArgs &a, int blocks, void *stream) {
  if (blocks <= 0) return 0;
  int per_wave = ((a.ix.stride * 4 + 15) & ~15) + 64 * 8 + 64 * 4 + 64 * 4 + ((a.k + 1) & ~1) * 8;
  per_wave = (per_wave + 15) & ~15;
  size_t lds = (size_t)per_wave * kWavesPerBlock;
  dim3 grid(blocks), block(64 * kWavesPerBlock);
  if (a.ix.metric == 1) hipLaunchKernelGGL(k_brute<1>, grid, block, lds, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(k_brute<0>, grid, block, lds, (hipStream_t)stream, a);
  return check(hipGetLastError());
}

int launch_finalize(const FinalizeArgs &a, void *stream) {
  if (a.nq == 0) return 0;
  int threads = 128;
  int blocks = (int)((a.nq + threads - 1) / threads);
  hipLaunchKernelGGL(k_finalize, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, a);
  return check(hipGetLastError());
}

}  // namespace wann
