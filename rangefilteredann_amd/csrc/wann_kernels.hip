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

#include <algorithm>

#include "wann_device.h"
#include "wann_wave.h"

namespace wann {

// --------------------------------------------------------------------------------------------
// k_search: one wavefront takes a task (query, partition) through the WHOLE post-filter loop of
// PostfilterVamanaIndex::query (postfilter_vamana.h:141-188): search at beam b, count the in-window
// beam entries, double b and search again from scratch while fewer than k were found, then the
// optional final re-search at min(b * final_beam_multiply, max_beam).  Doing the loop inside the
// wave (instead of one launch per doubling round) keeps the GPU busy while the few queries that
// need large beams run their long, strictly sequential searches.
//
// Per-wave LDS: common scratch + a pool of A.pool_bytes that holds, per search,
//   small: beam + seen-filter          (4 << bits) + 8 B <= pool
//   big  : beam only, filter in global (8 B <= pool)
//   huge : nothing (beam and filter in per-slot global scratch)
// Beams above A.cap_inkernel leave the kernel: speculative levels were routed to the companion launch
// (BIG = true) from the start, continuations go to its pollers, the rest (next_list / final_list) to
// follow-up launches of this same kernel.
// --------------------------------------------------------------------------------------------
__device__ __forceinline__ int hash_bits_dev(long long beam) {  // beamSearch.h:66
  long long sq = beam * beam;
  int e = (sq <= 1) ? 0 : (64 - __builtin_clzll((unsigned long long)(sq - 1)));  // ceil(log2(beam^2))
  e -= 2;
  return e < 10 ? 10 : e;
}

// BIG = false: the ordinary kernel (four independent waves per workgroup, each with its own LDS slice).
// BIG = true : the companion launch for speculative levels beyond cap_inkernel: ONE wave per workgroup that
// owns a large LDS pool (beams up to big_cap), runs concurrently with the ordinary kernel on a second stream,
// serves the static big list (longest class first) and then -- in its first npollers workgroups -- waits for
// continuations handed over by ordinary waves until every ordinary ticket is done.  Two kernels rather than
// two modes of one: the mode logic cost the ordinary kernel 20 VGPRs (it needs all 256 of two waves per SIMD).
template <int METRIC, bool BIG>
__global__ __launch_bounds__(BIG ? 128 : 64 * kWavesPerBlock) void k_search(SearchArgs A) {  // (!BIG must stay within 256 VGPRs: two waves per SIMD; an explicit occupancy hint made the schedule 5 % slower)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const IndexView &ix = A.ix;
  const int lane = lane_id();
  const int wib = threadIdx.x >> 6;
  // four searching waves per workgroup; the one-wave kernel has ONE searching wave (plus, with A.helper, its prefetch helper)
  const int slot = BIG ? (int)blockIdx.x : (int)(blockIdx.x * (blockDim.x >> 6) + wib);
  const int per_wave = wave_lds_common_bytes(ix.stride) + A.pool_bytes;
  unsigned char *base = smem + (BIG ? 0 : (size_t)wib * per_wave);
  // the helper's mailbox takes the last bytes of the pool
  PrefetchBox *const box = (BIG && A.helper) ? reinterpret_cast<PrefetchBox *>(base + per_wave - (int)sizeof(PrefetchBox)) : nullptr;
  if (BIG && A.helper) {
    if (wib == 0 && lane == 0) {
      volatile int32_t *vb = reinterpret_cast<volatile int32_t *>(box);
      for (int j = 0; j < (int)(sizeof(PrefetchBox) / 4); j++) vb[j] = 0;
    }
    __syncthreads();  // (the only workgroup barrier of the kernel: the mailbox is initialised before the helper polls it)
  }
  u64 *gbeam = (BIG && A.g_beam) ? A.g_beam + (size_t)slot * A.g_beam_cap : nullptr;  // (beams outside the LDS: one-wave kernel only)
  int32_t *const gtable = A.g_table ? A.g_table + ((size_t)slot << A.g_table_bits) : nullptr;
  const int heavy = A.heavy_count ? *A.heavy_count : 0;
  const int mid_end = heavy + (A.mid_count ? *A.mid_count : 0);
  const int total = mid_end + *A.list_count;  // ordinary tickets
  const int pool_bytes = A.pool_bytes - ((BIG && A.helper) ? (int)sizeof(PrefetchBox) : 0), cap = A.cap_inkernel;
  // BIG: the first npollers workgroups only serve continuations, so that one starts as soon as it is handed over
  bool polling = BIG && A.big_list && (int)blockIdx.x < A.npollers;

  // ordinary launch with a companion: as many ordinary workgroups as there are big items (+ pollers) leave at once, so that
  // the companion's workgroups (launched first, but the LDS of every CU is fully booked by this launch) find room
  if (!BIG && A.yield_for_big) {
    const int items = A.big_count[0] + A.big_count[1];
    if ((int)blockIdx.x < min(items + A.npollers, (int)gridDim.x / 2)) return;
  }

  if (BIG && A.helper && wib == 1) {  // the prefetch helper wave of this workgroup's search wave
    prefetch_helper(ix, A.g_table ? A.g_table + ((size_t)slot << A.g_table_bits) : nullptr,
                    A.g_seen ? A.g_seen + (size_t)slot * A.g_seen_words : nullptr, A.degree_limit, box);
    return;
  }
  for (;;) {
    int ti;
    bool dyn = false;
    if (BIG && !polling && A.big_list) {
      const int big_first = A.big_count[0], big_total = big_first + A.big_count[1];
      const int t = wave_ticket(A.big_cursor);
      if (t >= big_total) break;
      ti = (t < big_first) ? A.big_list[t] : A.big_list[A.big_stride + t - big_first];
    } else if (BIG && polling) {
      // wait for continuation number d, or for the end of all ordinary work (a producer publishes its item
      // before it reports its ticket done, so the item count is final once done_count == total)
      const int d = wave_ticket(A.dyn_cursor);
      int item = -1;
      bool settled = false;
      // Bounded wait.  The producers are waves of a DIFFERENT launch; HIP does not promise that the two launches are
      // resident together (a shared hardware queue, HIP_LAUNCH_BLOCKING / AMD_SERIALIZE_KERNEL, rocprofv3 --pmc all
      // serialise them).  A poller gives up soon when no ordinary wave has ever taken a ticket (the other launch has
      // not started: it may be queued BEHIND this one) and after a long bound otherwise; whatever it leaves unserved
      // stays in dyn_list (entries >= 0) and the host hands it to a follow-up launch (run_batch).
      const int spins = A.force_poll_timeout ? 0 : (1 << 22);
      for (int spin = 0; spin < spins; spin++) {
        int have = 0, fin = 0, idle = 0;
        if (lane == 0) {
          // relaxed device-scope atomics: served by the memory side, no cache invalidation per poll
          fin = __hip_atomic_load(A.done_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= total;
          have = __hip_atomic_load(A.dyn_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > d;
          if (have) item = __hip_atomic_load(A.dyn_list + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (spin >= (1 << 12) && (spin & 255) == 0)
            idle = total > 0 && __hip_atomic_load(A.cursor, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;
        }
        item = uni(item);
        if (item >= 0 || (uni(fin) && !uni(have))) {
          settled = true;
          break;
        }
        if (uni(idle)) break;  // serialised dispatch: nobody will produce while this launch occupies the queue
        __builtin_amdgcn_s_sleep(32);
      }
      if (!settled && lane == 0) atomicAdd(&A.ctr->poll_timeouts, 1ull);
      if (item < 0) break;  // nothing more can arrive (or given up)
      if (lane == 0) __hip_atomic_store(A.dyn_list + d, -2 - item, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // served
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // the producer's writes (next_beam) before its publication
      ti = item;
      dyn = true;
    } else {
      const int t = wave_ticket(A.cursor);
      if (t >= total) break;
      ti = (t < heavy) ? A.heavy_list[t] : (t < mid_end) ? A.mid_list[t - heavy] : A.list[t - mid_end];  // long searches start first
    }
    // (the task's fields are read where they are used, not kept in scalar registers across the searches)
    const PartDesc part = ix.parts[A.tasks[ti].part];
    const int64_t qrow = A.tasks[ti].query;
    const int64_t qid = A.raw ? A.raw_qids[qrow] : (A.qid_base + qrow);
    const int64_t row_off = part.start;

    long long b = A.B;
    bool final_pass = A.is_final != 0;
    bool sub = (A.tasks[ti].flags & 4) != 0;  // speculative sub-task: ONE search at beam B << level
    if (BIG && dyn) {  // a continuation: a plain task or a resolved parent, at the beam its producer recorded
      sub = false;
      b = __hip_atomic_load(A.next_beam + ti, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (sub) b = (long long)A.B << (int)A.tasks[ti].a;
    else if (A.start_beam) b = A.start_beam[ti];
    for (;;) {  // postfilter_vamana.h:161-181
      const int B = (int)b;
      const int bits = hash_bits_dev(b);
      const int beam_bytes = ((B + 1) & ~1) * 8;
      const bool beam_lds = beam_bytes <= pool_bytes;
      const bool table_lds = beam_lds && (beam_bytes + (4 << bits) <= pool_bytes);
      WaveLds L = carve_wave_lds(base, ix.stride, B, beam_lds);
      // stage the query (zero padded); every search restarts from scratch
      for (int i = lane; i < ix.stride; i += 64) L.qv[i] = (i < ix.d) ? A.queries[qrow * ix.d + i] : 0.f;
      WAVE_SYNC();
      int m;
      long long nvis, ncmp;
#ifdef WANN_TASK_TRACE
      const long long trace_t0 = A.trace ? (long long)wall_clock64() : 0;
#endif
      // the part of the pool that the beam leaves free serves as the clash-detection scratch of the general cores
      int32_t *mini = nullptr;
      uint32_t mini_mask = 0;
      const int free_words = beam_lds ? (pool_bytes - beam_bytes) >> 2 : 0;
      if (free_words >= 1024 && !A.force_general) {
        mini = reinterpret_cast<int32_t *>(reinterpret_cast<unsigned char *>(L.lbeam) + beam_bytes);
        mini_mask = (1u << (31 - __builtin_clz((unsigned)free_words))) - 1u;
      }
      const bool small_ok = table_lds && !A.force_general;
      if (small_ok && B <= 64)
        wave_beam_search_small<METRIC, 1>(ix, part, L, B, bits, qid, A.limit, A.degree_limit, m, nvis, ncmp, A.prof);
      else if (small_ok && B <= 128)
        wave_beam_search_small<METRIC, 2>(ix, part, L, B, bits, qid, A.limit, A.degree_limit, m, nvis, ncmp, A.prof);
      else if (!BIG && beam_lds && gtable) {
        // Four-wave kernel, beams 129 .. cap_inkernel: the first-generation general core.  (Measured: the second-
        // generation core below wins from beams of about 2 000 entries on -- the one-wave kernel's range -- and loses
        // a tenth per hop at beams of a few hundred, where many candidates pass per hop.)
        wave_beam_search<METRIC, false, true, false>(ix, part, L, nullptr, gtable, B, bits, qid, A.limit, A.degree_limit,
                                                     nullptr, 0, m, nvis, ncmp, A.prof, mini, mini_mask);
      } else if (BIG && beam_lds && A.g_seen && !A.old_general) {
        // Second-generation general core.  Tagged filter entries: this search takes the slot's next epoch; on
        // wrap-around (or after a search that stored plain ids) the slot's whole region is zeroed.  Partitions of
        // more than 2^24 nodes use plain ids and clear what they use.
        int e = 0;
        if (lane == 0) e = A.g_epoch[slot];
        e = uni(e);
        uint32_t tag = 0;
        if (part.n <= (1 << 24)) {
          if (e >= 254) {
            int4 *gt = reinterpret_cast<int4 *>(gtable);
            for (int i = lane; i < (1 << (A.g_table_bits - 2)); i += 64) gt[i] = make_int4(0, 0, 0, 0);
            e = 0;
          }
          e++;
          tag = (uint32_t)e << 24;
        } else {
          int4 *gt = reinterpret_cast<int4 *>(gtable);
          for (int i = lane; i < (1 << (bits - 2)); i += 64) gt[i] = make_int4(-1, -1, -1, -1);
          e = 254;
        }
        if (lane == 0) A.g_epoch[slot] = e;
        if (!mini) {  // no room beside the beam: the merge scratch (unused during the filter step) serves
          mini = reinterpret_cast<int32_t *>(L.cand_key);
          mini_mask = 127u;
        }
        wave_beam_search_big<METRIC>(ix, part, L, gtable, tag, A.g_seen + (size_t)slot * A.g_seen_words, B, bits, qid, A.limit,
                                     A.degree_limit, mini, mini_mask, m, nvis, ncmp, A.prof, box, A.tasks[ti].part);
      } else if (BIG) {
        // First-generation general cores: only in the one-wave-per-workgroup kernel (512 registers per wave), which
        // serves the companion launch, the follow-up launches and the test / dev switches.
        if (A.g_epoch && lane == 0) A.g_epoch[slot] = 254;  // plain ids go into the table: the next tagged search clears it
        uint32_t *const vset = (A.cut_k > 0 && A.g_seen) ? A.g_seen + (size_t)slot * A.g_seen_words : nullptr;
        if (A.cut_k > 0 && beam_lds)  // unfiltered VamanaIndex queries: the k / cut step (global filter: the host forces one)
          wave_beam_search<METRIC, false, true, false, true>(ix, part, L, nullptr, gtable, B, bits, qid, A.limit, A.degree_limit,
                                                             nullptr, 0, m, nvis, ncmp, A.prof, mini, mini_mask, A.cut_k, A.cut, vset);
        else if (table_lds)
          wave_beam_search<METRIC, true, true, false>(ix, part, L, nullptr, nullptr, B, bits, qid, A.limit, A.degree_limit,
                                                      nullptr, 0, m, nvis, ncmp, A.prof);
        else if (beam_lds)
          wave_beam_search<METRIC, false, true, false>(ix, part, L, nullptr, gtable, B, bits, qid, A.limit, A.degree_limit,
                                                       nullptr, 0, m, nvis, ncmp, A.prof, mini, mini_mask);
        else
          wave_beam_search<METRIC, false, false, false>(ix, part, L, gbeam, gtable, B, bits, qid, A.limit, A.degree_limit,
                                                        nullptr, 0, m, nvis, ncmp, A.prof);
      } else {
        // (the host never gives the four-wave kernel a beam that needs one of those: see config_for)
        m = 0;
        nvis = ncmp = 0;
        if (lane == 0) atomicAdd(&A.ctr->unsupported, 1ull);
      }
      auto beam_ld = [&](int i) -> u64 { return (!BIG || beam_lds) ? L.lbeam[i] : gbeam[i]; };
#ifdef WANN_TASK_TRACE
      if (A.trace && lane == 0) {
        long long *rec = A.trace + 1 + 4 * atomicAdd((unsigned long long *)A.trace, 1ull);
        rec[0] = ti | (sub ? 1ll << 40 : 0) | (BIG ? 1ll << 41 : 0);
        rec[1] = B;
        rec[2] = trace_t0;
        rec[3] = (long long)wall_clock64();
      }
#endif
      if (lane == 0) {
        if (sub) {  // attributed when the parent is resolved
          A.sub_hops[ti] = nvis;
          A.sub_cmps[ti] = ncmp;
        } else {
          atomicAdd(&A.ctr->beam_searches, 1ull);
          atomicAdd(&A.ctr->hops, (unsigned long long)nvis);
          atomicAdd(&A.ctr->dist_cmps, (unsigned long long)ncmp);
        }
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
        break;
      }
      // ---- post filter: keep beam entries whose label lies in [lo,hi], first k of them
      //      (postfilter_vamana.h:234-251); ids become sorted-order indices (subset[local])
      int found = 0;
      long long labs = 0;
      const float win_lo = A.tasks[ti].lo, win_hi = A.tasks[ti].hi;
      for (int bx = 0; bx < m && found < A.k; bx += 64) {
        int x = bx + lane;
        bool act = x < m;
        u64 e = act ? beam_ld(x) : 0ull;
        int lid = (int)((uint32_t)e >> 1);
        float lab = act ? ix.labels[row_off + lid] : 0.f;
        bool inw = act && (lab >= win_lo) && (lab <= win_hi);
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
      }
      WAVE_SYNC();
      if (sub) {
        // Publish this level, then the LAST sub-task to finish replays the sequential rule over the
        // levels: the result is that of the first level with >= k in-window entries (or of the last
        // level), exactly what the doubling loop returns; it then continues as the parent.
        const int parent = (int)A.tasks[ti].b;
        __threadfence();
        int old_done = 0;
        if (lane == 0) old_done = atomicAdd(&A.par_done[parent], 1);
        old_done = uni(old_done);
        const int nsub = (int)A.tasks[parent].a, sbase = (int)A.tasks[parent].b;
        if (old_done + 1 != nsub) break;  // not the last one: take the next ticket
        __threadfence();
        int succ = -1;
        for (int r = 0; r < nsub; r++)
          if (__hip_atomic_load(&A.out_cnt[sbase + r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= A.k) {
            succ = r;
            break;
          }
        const int upto = succ >= 0 ? succ : nsub - 1;
        if (lane == 0) {
          unsigned long long h0 = 0, c0 = 0, h1 = 0, c1 = 0;
          for (int r = 0; r < nsub; r++) {
            const unsigned long long hh = (unsigned long long)__hip_atomic_load(&A.sub_hops[sbase + r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long cc = (unsigned long long)__hip_atomic_load(&A.sub_cmps[sbase + r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (r <= upto) {
              h0 += hh;
              c0 += cc;
            } else {
              h1 += hh;
              c1 += cc;
            }
          }
          atomicAdd(&A.ctr->beam_searches, (unsigned long long)(upto + 1));
          atomicAdd(&A.ctr->hops, h0);
          atomicAdd(&A.ctr->dist_cmps, c0);
          atomicAdd(&A.ctr->spec_searches, (unsigned long long)(nsub - 1 - upto));
          atomicAdd(&A.ctr->spec_hops, h1);
          atomicAdd(&A.ctr->spec_dist_cmps, c1);
        }
        found = __hip_atomic_load(&A.out_cnt[sbase + upto], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int j = lane; j < found; j += 64)
          A.out_key[(size_t)parent * A.k + j] =
              __hip_atomic_load(&A.out_key[(size_t)(sbase + upto) * A.k + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (lane == 0) A.out_cnt[parent] = found;
        WAVE_SYNC();
        ti = parent;
        sub = false;
        b = (long long)A.B << upto;
      }
      if (final_pass) break;
      if (found >= A.k) {  // doubling loop ends here (postfilter_vamana.h:161-172); final re-search?
        long long fb = b * ((A.tasks[ti].flags & 2) ? 1 : A.mult);
        if (fb > A.max_beam) fb = A.max_beam;
        if (fb <= b) break;
        if (fb <= cap) {
          b = fb;
          final_pass = true;
          continue;
        }
        if (lane == 0) {
          int at = atomicAdd(A.final_count, 1);
          A.final_list[at] = ti;
          A.next_beam[ti] = (int32_t)fb;
        }
        break;
      }
      const long long nb = 2 * b;
      if (nb >= A.max_beam) break;  // cannot double any more: the short result stands
      if (nb > cap) {
        if (lane == 0) {
          if (A.next_beam) A.next_beam[ti] = (int32_t)nb;
          if (!BIG && A.npollers > 0 && nb <= A.big_cap) {  // to a poller of the companion launch
            __threadfence();
            const int d = atomicAdd(A.dyn_count, 1);
            __hip_atomic_store(A.dyn_list + d, ti, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
          } else {
            A.next_list[atomicAdd(A.next_count, 1)] = ti;
          }
        }
        break;
      }
      b = nb;
    }
    if (!BIG && A.done_count) {
      WAVE_SYNC();
      // relaxed: a continuation is published by an atomic whose result this wave has already waited for, so it
      // is counted in dyn_count before this ticket is counted as done; no cache write-back per ticket
      if (lane == 0) __hip_atomic_fetch_add(A.done_count, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (BIG && box && lane == 0) *reinterpret_cast<volatile int32_t *>(box) = -1;  // the helper wave leaves too
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
  // scans are short and uniform: a wave takes four tickets at a time and reports its row count once (ten thousand
  // same-address atomics per batch were most of a tiny-window batch)
  constexpr int kTickets = 4;
  unsigned long long rows_done = 0;

  for (int t = 0, tend = 0;; t++) {
    if (t >= tend) {
      int lane0;
      asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane0));  // (see wave_ticket)
      int tt = 0;
      if (lane0 == 0) tt = atomicAdd(A.cursor, kTickets);
      t = __builtin_amdgcn_readfirstlane(tt);
      tend = t + kTickets;
    }
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
    if (lane == 0) A.out_cnt[ti] = m;
    rows_done += (unsigned long long)(task.b > task.a ? task.b - task.a : 0);
    WAVE_SYNC();
  }
  if (lane == 0 && rows_done) atomicAdd(&A.ctr->brute_rows, rows_done);
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

// the two lower bounds of a window with their probes issued together: two independent chains of ~log2(n)
// dependent loads cost the latency of one
__device__ __forceinline__ void first_ge2(const float *fv, int64_t n, float v1, float v2, uint64_t &r1, uint64_t &r2) {
  const float f0 = fv[0];
  int64_t s1 = 0, e1 = n, s2 = 0, e2 = n;
  if (f0 >= v1) e1 = 0;  // (tree_utils.h:20-22: index 0 is special-cased)
  if (f0 >= v2) e2 = 0;
  while (s1 + 1 < e1 || s2 + 1 < e2) {
    const bool g1 = s1 + 1 < e1, g2 = s2 + 1 < e2;
    const int64_t m1 = (s1 + e1) / 2, m2 = (s2 + e2) / 2;
    const float x1 = g1 ? fv[m1] : 0.f, x2 = g2 ? fv[m2] : 0.f;
    if (g1) {
      if (x1 >= v1) e1 = m1;
      else s1 = m1;
    }
    if (g2) {
      if (x2 >= v2) e2 = m2;
      else s2 = m2;
    }
  }
  r1 = (uint64_t)e1;
  r2 = (uint64_t)e2;
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

// ---- task emission: a query owns the slots tasks[q*maxt .. q*maxt+maxt) ---------------------------
struct Emitter {
  const RouteArgs &A;
  int64_t q;
  int n;
  __device__ Emitter(const RouteArgs &a, int64_t qq) : A(a), q(qq), n(0) {}
  __device__ __forceinline__ void push(const Task &t) {
    if (t.mode == T_EMPTY) return;
    if (n >= A.maxt) {  // the host raises an error for the batch
      atomicAdd(&A.ctr->unsupported, 1ull);
      return;
    }
    const int32_t ti = (int32_t)(q * A.maxt + n);
    n++;
    A.tasks[ti] = t;
    if (t.mode == T_GRAPH) {
      if (t.flags & 1) A.heavy_list[atomicAdd(A.heavy_count, 1)] = ti;
      else if (t.flags & 8) A.mid_list[atomicAdd(A.mid_count, 1)] = ti;
      else A.graph_list[atomicAdd(A.graph_count, 1)] = ti;
    } else {
      A.brute_list[atomicAdd(A.brute_count, 1)] = ti;
    }
  }
  // SpatialIndex::query on partition pidx for window [lo,hi]: the post-filter loop on a Vamana
  // leaf (postfilter_vamana.h:141-188), brute force on a PrefilterIndex leaf (prefiltering.h:154-204)
  __device__ __forceinline__ void leaf(int32_t pidx, float lo, float hi, uint64_t w, bool mult_one) {
    const IndexView &ix = A.ix;
    Task t;
    t.query = (int32_t)q;
    t.part = pidx;
    t.flags = 0;
    t.a = t.b = 0;
    t.lo = lo;
    t.hi = hi;
    const PartDesc pd = ix.parts[pidx];
    if (ix.vamana_leaves) {
      t.mode = (A.beam < A.max_beam) ? T_GRAPH : T_EMPTY;  // postfilter_vamana.h:161: no search otherwise
      if (mult_one) t.flags |= 2;  // three_split centre: final_beam_multiply forced to 1
      // A window that is a small fraction of its partition needs several doublings, i.e. a long chain of
      // strictly sequential searches.  Such a task is started first and its doubling levels are searched
      // CONCURRENTLY by different waves (each level restarts from scratch anyway, postfilter_vamana.h:
      // 161-172); the sequential rule "first level with >= k in-window results" is applied afterwards.
      // a first beam that expects fewer than 4k in-window entries fails now and then: such a task starts before the
      // bulk, so that its second, longer search is not what the launch ends with
      if (t.mode == T_GRAPH && (uint64_t)A.beam * w < 4ull * (uint64_t)A.k * (uint64_t)pd.n) t.flags |= 8;
      if (t.mode == T_GRAPH && w > 0 && 2ull * (uint64_t)A.k * (uint64_t)pd.n >= (uint64_t)A.cap_inkernel * w) atomicAdd(A.risk_count, 1);
      if (t.mode == T_GRAPH && w > 0 && (uint64_t)pd.n / w >= (uint64_t)A.heavy_ratio) {
        t.flags |= 1;
        if (A.spec && n < A.maxt) {
          // expected in-window share of a beam ~ w / partition size: levels up to the first beam with
          // beam * w / n_p >= k, plus one
          int nsub = 0;
          long long bb = A.beam;
          const unsigned long long need = (unsigned long long)A.k * (uint64_t)pd.n;
          while (bb < A.max_beam && (bb <= A.cap_inkernel || bb <= A.big_cap) && nsub < 12) {
            nsub++;
            if ((unsigned long long)bb * w >= need * (unsigned long long)A.spec_num / 8ull) break;
            bb *= 2;
          }
          if (nsub >= 2) {
            const int base = atomicAdd(A.sub_count, nsub);
            if (A.sub_base0 + base + nsub <= A.sub_cap) {
              const int32_t pti = (int32_t)(q * A.maxt + n);
              Task parent = t;
              parent.mode = T_PARENT;
              parent.a = nsub;
              parent.b = A.sub_base0 + base;
              A.tasks[pti] = parent;
              n++;
              for (int r = nsub - 1; r >= 0; r--) {  // longest search first
                Task st = t;
                st.flags |= 4;
                st.a = r;
                st.b = pti;
                const int32_t sti = A.sub_base0 + base + r;
                A.tasks[sti] = st;
                if (((long long)A.beam << r) > A.cap_inkernel) {
                  const int cls = (((long long)A.beam << r) >= 4096) ? 0 : 1;
                  A.big_list[cls * A.big_stride + atomicAdd(A.big_count + cls, 1)] = sti;
                }
                else A.heavy_list[atomicAdd(A.heavy_count, 1)] = sti;
              }
              return;
            }
          }
        }
      }
    } else {
      const int64_t s = prefilter_bound(ix.labels + pd.start, pd.n, lo);
      const int64_t e = prefilter_bound(ix.labels + pd.start, pd.n, hi);
      t.a = pd.start + s;
      t.b = pd.start + e;
      t.mode = (e > s) ? T_BRUTE : T_EMPTY;
    }
    push(t);
  }
  __device__ __forceinline__ void brute(uint64_t a, uint64_t b) {  // rows [a,b) of the sorted order, no label test
    if (b <= a) return;
    Task t;
    t.query = (int32_t)q;
    t.mode = T_BRUTE;
    t.part = 0;
    t.flags = 0;
    t.a = (int64_t)a;
    t.b = (int64_t)b;
    t.lo = t.hi = 0.f;
    push(t);
  }
};

struct Centre {
  int64_t row, first, last;
  uint64_t cover_start, cover_end;
};

__device__ __forceinline__ int64_t bucket_containing(const int64_t *off, int64_t nb, uint64_t index) {
  int64_t lo = 0, hi = nb;  // largest b with off[b] <= index   (range_filter_tree.h:213-232)
  while (lo + 1 < hi) {
    const int64_t mid = (lo + hi) / 2;
    if ((uint64_t)off[mid] <= index) lo = mid;
    else hi = mid;
  }
  return lo;
}

// find_largest_ranges_within_query_range (range_filter_tree.h:234-295)
__device__ __forceinline__ bool find_centre(const IndexView &ix, uint64_t istart, uint64_t eend, Centre &c) {
  const uint64_t range_size = eend - istart;
  int64_t row = -1;
  for (int r = 0; r < ix.nlevels; r++) {
    const int64_t *off = ix.wst_off + ix.wst_ptr[r];
    if ((uint64_t)(off[1] - off[0] - 1) <= range_size) {
      row = r;
      break;
    }
  }
  if (row < 0) return false;
  const int64_t *off = ix.wst_off + ix.wst_ptr[row];
  int64_t nb = ix.level_nb[row];
  int64_t first = (istart == 0) ? 0 : bucket_containing(off, nb, istart - 1) + 1;
  if (first >= nb) return false;  // the reference indexes past the row here (out_of_range)
  uint64_t start = (uint64_t)off[first], end = (uint64_t)off[first + 1];
  if (end > eend) {
    row += 1;
    if (row >= ix.nlevels) return false;
    off = ix.wst_off + ix.wst_ptr[row];
    nb = ix.level_nb[row];
    first = (istart == 0) ? 0 : bucket_containing(off, nb, istart - 1) + 1;
    if (first >= nb) return false;
    start = (uint64_t)off[first];
    end = (uint64_t)off[first + 1];
  }
  int64_t last = first + 1;
  while (last < nb) {
    const uint64_t next_end = (uint64_t)off[last + 1];
    if (next_end > eend) break;
    last++;
    end = next_end;
  }
  c.row = row;
  c.first = first;
  c.last = last;
  c.cover_start = start;
  c.cover_end = end;
  return true;
}

// Tree query methods (range_filter_tree.h:62-96).  The reference's three entry points call each other:
// three_split_search (:473-540) covers its two remainders with optimized_postfiltering_search (:403-471),
// which falls back to fenwick_tree_search (:297-401) for tiny windows / a large bucket-to-window ratio.
// Here one loop walks the (at most three) label windows a query decomposes into and each body exists
// once, fully inlined: k_route must not need a scratch segment (no calls, no stack objects).
enum { W_FENWICK = 0, W_OPTIMIZED = 1, W_THREE_SPLIT = 2 };

__device__ __forceinline__ void emit_tree(Emitter &E, float lo0, float hi0, int mode0) {
  const IndexView &ix = E.A.ix;
  float lo1 = 0.f, hi1 = 0.f, lo2 = 0.f, hi2 = 0.f;  // remainders of three_split (always W_OPTIMIZED)
  int nwin = 1;
  for (int it = 0; it < nwin; it++) {
    const float lo = it == 0 ? lo0 : (it == 1 ? lo1 : lo2);
    const float hi = it == 0 ? hi0 : (it == 1 ? hi1 : hi2);
    int mode = it == 0 ? mode0 : W_OPTIMIZED;
    bool mult_one = false;
    if (hi < ix.labels[0] || lo > ix.labels[ix.n - 1]) continue;  // check_empty (:191-203)
    uint64_t istart, eend;
    first_ge2(ix.labels, ix.n, lo, hi, istart, eend);
    const uint64_t w = eend - istart;
    Centre c;
    bool have_centre = false;
    if (mode != W_OPTIMIZED) have_centre = find_centre(ix, istart, eend, c);

    if (mode == W_THREE_SPLIT) {  // three_split_search (:473-540)
      if (have_centre) {
        for (int64_t b = c.first; b < c.last; b++) E.leaf((int32_t)(ix.level_part0[c.row] + b), lo, hi, w, true);
        if (c.cover_start - istart > 0) {
          lo1 = lo;
          hi1 = ix.labels[c.cover_start];
          nwin = 2;
        }
        if (eend - c.cover_end > 0) {
          if (nwin == 2) {
            lo2 = ix.labels[c.cover_end];
            hi2 = hi;
          } else {
            lo1 = ix.labels[c.cover_end];
            hi1 = hi;
          }
          nwin++;
        }
        continue;
      }
      mode = W_FENWICK;  // qp_fenwick: final_beam_multiply = 1 (:490-498)
      mult_one = true;
    }

    if (mode == W_OPTIMIZED) {  // optimized_postfiltering_search (:403-471)
      bool fallback = 4 * w < (uint64_t)(int64_t)ix.cutoff;  // :419-421
      if (!fallback) {
        int64_t row = 0, idx = 0;
        while (row + 1 < ix.nlevels) {  // :426-451
          const int64_t nrow = row + 1;
          const int64_t *off = ix.wst_off + ix.wst_ptr[nrow];
          int64_t nidx = -1;
          for (int64_t ch = idx * ix.split; ch < idx * ix.split + ix.split; ch++) {
            if (ch >= ix.level_nb[nrow]) break;
            if (istart >= (uint64_t)off[ch] && eend <= (uint64_t)off[ch + 1]) nidx = ch;
          }
          if (nidx < 0) break;
          idx = nidx;
          row = nrow;
        }
        if (E.A.has_ratio) {  // :460-466
          const int64_t *off = ix.wst_off + ix.wst_ptr[row];
          const float ratio = (float)(uint64_t)(off[idx + 1] - off[idx]) / (float)w;
          fallback = ratio > E.A.ratio;
        }
        if (!fallback) {
          E.leaf((int32_t)(ix.level_part0[row] + idx), lo, hi, w, false);
          continue;
        }
      }
      mode = W_FENWICK;
      have_centre = find_centre(ix, istart, eend, c);
    }

    // fenwick_tree_search (:297-401)
    if (!have_centre) {
      E.brute(istart, eend);
      continue;
    }
    for (int64_t b = c.first; b < c.last; b++) E.leaf((int32_t)(ix.level_part0[c.row] + b), lo, hi, w, mult_one);
    uint64_t cov_s = c.cover_start, cov_e = c.cover_end;
    int64_t left = c.first, right = c.last - 1;
    const int64_t B = ix.split;
    for (int64_t row = c.row + 1; row < ix.nlevels; row++) {
      const int64_t *off = ix.wst_off + ix.wst_ptr[row];
      const int64_t nb = ix.level_nb[row];
      left *= B;
      right = right * B + B - 1;
      while (left > 0) {
        const uint64_t nls = (uint64_t)off[left - 1];
        if (nls < istart) break;
        cov_s = nls;
        left -= 1;
        E.leaf((int32_t)(ix.level_part0[row] + left), lo, hi, w, mult_one);
      }
      while (right < nb - 1) {
        const uint64_t nre = (uint64_t)off[right + 2];
        if (nre > eend) break;
        cov_e = nre;
        right += 1;
        E.leaf((int32_t)(ix.level_part0[row] + right), lo, hi, w, mult_one);
      }
    }
    E.brute(istart, cov_s);
    E.brute(cov_e, eend);
  }
}

// super_optimized_postfiltering_search (super_optimized_postfilter_tree.h:187-270)
__device__ __forceinline__ void emit_super(Emitter &E, float lo, float hi) {
  const IndexView &ix = E.A.ix;
  if (hi < ix.labels[0] || lo > ix.labels[ix.n - 1]) return;
  uint64_t istart, eend;
  first_ge2(ix.labels, ix.n, lo, hi, istart, eend);
  const uint64_t w = eend - istart;
  int level;
  int64_t idx = 0;
  for (level = ix.nlevels - 1; level >= 0; level--) {
    if (level == 0) {
      idx = 0;
      break;
    }
    const uint64_t bsz = (uint64_t)ix.sup_size[level];
    if (bsz < w) continue;
    const uint64_t shift = (uint64_t)ix.sup_shift[level];
    const uint64_t nb = (uint64_t)ix.level_nb[level];
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
  E.leaf((int32_t)(ix.level_part0[level] + idx), lo, hi, w, false);
}

// KIND is the index class (wann.h WANN_KIND_*; 2 stands for both tree kinds): one small kernel per class.
template <int KIND>
__global__ void k_route(RouteArgs A) {
  const IndexView &ix = A.ix;
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= A.nq) return;
  const float lo = A.ranges[2 * q], hi = A.ranges[2 * q + 1];
  Emitter E(A, q);
  if (KIND == 0) {  // PrefilterIndex: [lb(lo), lb(hi)) over the label argsort (prefiltering.h:159-184)
    Task t;
    t.query = (int32_t)q;
    t.part = 0;
    t.flags = 0;
    t.lo = lo;
    t.hi = hi;
    t.a = prefilter_bound(ix.fv_sorted, ix.n, lo);
    t.b = prefilter_bound(ix.fv_sorted, ix.n, hi);
    t.mode = (t.b > t.a) ? T_BRUTE_GATHER : T_EMPTY;
    E.push(t);
  } else if (KIND == 1) {  // stand-alone PostfilterVamanaIndex: always the one graph, no window lookup
    Task t;
    t.query = (int32_t)q;
    t.part = 0;
    t.flags = 0;
    t.a = t.b = 0;
    t.lo = lo;
    t.hi = hi;
    t.mode = (A.beam < A.max_beam) ? T_GRAPH : T_EMPTY;
    E.push(t);
  } else if (KIND == 4) {
    emit_super(E, lo, hi);
  } else {
    emit_tree(E, lo, hi, A.method == M_OPTIMIZED ? W_OPTIMIZED : (A.method == M_THREE_SPLIT ? W_THREE_SPLIT : W_FENWICK));
  }
  A.qtask_cnt[q] = E.n;
}

// One thread per query when every query has at most one task; the multi-task form (fenwick,
// three_split) merges the per-task top-k lists: concatenate, sort by (dist, id), truncate
// (range_filter_tree.h:542-549; duplicates are kept like the reference keeps them).
__global__ void k_finalize(FinalizeArgs A) {
  int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= A.nq) return;
  const int nt = A.qtask_cnt[q];
  const int64_t ti = q * A.maxt;
  const int cnt = nt > 0 ? A.out_cnt[ti] : 0;
  const bool decode = A.decode && (nt == 0 || A.tasks[ti].mode != T_BRUTE_GATHER);
  for (int j = 0; j < A.k; j++) {
    uint32_t id = A.pad_id;
    float dist = 3.402823466e+38f;  // std::numeric_limits<float>::max()
    if (j < cnt) {
      u64 e = A.out_key[(size_t)ti * A.k + j];
      id = (uint32_t)e;
      if (decode) id = A.ix.decoding[id];
      dist = funkey((uint32_t)(e >> 32));
    }
    A.ids[q * A.k + j] = id;
    A.dists[q * A.k + j] = dist;
  }
}

__global__ __launch_bounds__(64 * kWavesPerBlock) void k_finalize_multi(FinalizeArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = lane_id();
  const int wib = threadIdx.x >> 6;
  const int K = A.k;
  const int per_wave = (64 * 8 + ((K + 1) & ~1) * 8 + 15) & ~15;
  u64 *cand_key = reinterpret_cast<u64 *>(smem + (size_t)wib * per_wave);
  u64 *top = cand_key + 64;
  for (int64_t q = (int64_t)blockIdx.x * kWavesPerBlock + wib; q < A.nq; q += (int64_t)gridDim.x * kWavesPerBlock) {
    const int nt = A.qtask_cnt[q];
    int m = 0;
    for (int t = 0; t < nt; t++) {
      const int64_t ti = q * A.maxt + t;
      const int cnt = A.out_cnt[ti];
      for (int c0 = 0; c0 < cnt; c0 += 64) {
        const bool have = (c0 + lane) < cnt;
        u64 e = have ? A.out_key[(size_t)ti * K + c0 + lane] : 0ull;
        // out_key = fkey(dist) << 32 | sorted id  ->  merge key with the id shifted (bit 0 = flag)
        u64 key = (e & 0xffffffff00000000ull) | ((u64)(uint32_t)e << 1);
        bool pass = have;
        if (pass && m >= K) pass = (key | 1ull) < (top[K - 1] | 1ull);
        int p0;
        m = wave_merge<u64 *, false>(top, m, K, pass, key, cand_key, &p0);
      }
    }
    for (int j = lane; j < K; j += 64) {
      uint32_t id = A.pad_id;
      float dist = 3.402823466e+38f;
      if (j < m) {
        const u64 e = top[j];
        id = (uint32_t)e >> 1;
        if (A.decode) id = A.ix.decoding[id];
        dist = funkey((uint32_t)(e >> 32));
      }
      A.ids[q * K + j] = id;
      A.dists[q * K + j] = dist;
    }
    WAVE_SYNC();
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

int search_lds_bytes_per_wave(int stride, int pool_bytes) {
  return ((stride * 4 + 15) & ~15) + 64 * 8 + 64 * 4 + 64 * 4 + pool_bytes;
}

int launch_route(const RouteArgs &a, void *stream) {
  if (a.nq == 0) return 0;
  int threads = 128;
  int blocks = (int)((a.nq + threads - 1) / threads);
  hipStream_t s = (hipStream_t)stream;
  switch (a.ix.kind) {
    case 0: hipLaunchKernelGGL(k_route<0>, dim3(blocks), dim3(threads), 0, s, a); break;
    case 1: hipLaunchKernelGGL(k_route<1>, dim3(blocks), dim3(threads), 0, s, a); break;
    case 4: hipLaunchKernelGGL(k_route<4>, dim3(blocks), dim3(threads), 0, s, a); break;
    default: hipLaunchKernelGGL(k_route<2>, dim3(blocks), dim3(threads), 0, s, a); break;
  }
  return check(hipGetLastError());
}

template <int METRIC, bool BIG>
static int launch_search_t(const SearchArgs &a, dim3 grid, dim3 block, size_t lds, hipStream_t s) {
  auto kern = k_search<METRIC, BIG>;
  if (lds > 48 * 1024)
    if (check(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds))) return 1;
  hipLaunchKernelGGL(kern, grid, block, lds, s, a);
  return check(hipGetLastError());
}

int launch_search(const SearchArgs &a, const LaunchCfg &cfg, void *stream) {
  if (cfg.blocks <= 0) return 0;
  const int wpb = cfg.waves_per_block > 0 ? cfg.waves_per_block : kWavesPerBlock;
  size_t lds = (size_t)search_lds_bytes_per_wave(a.ix.stride, a.pool_bytes) * wpb;
  dim3 grid(cfg.blocks), block(64 * wpb * ((cfg.big && a.helper) ? 2 : 1));  // (+ the prefetch helper wave)
  hipStream_t s = (hipStream_t)stream;
  if (cfg.big) return a.ix.metric == 1 ? launch_search_t<1, true>(a, grid, block, lds, s) : launch_search_t<0, true>(a, grid, block, lds, s);
  return a.ix.metric == 1 ? launch_search_t<1, false>(a, grid, block, lds, s) : launch_search_t<0, false>(a, grid, block, lds, s);
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
  if (a.maxt > 1) {
    int per_wave = (64 * 8 + ((a.k + 1) & ~1) * 8 + 15) & ~15;
    int blocks = (int)std::min<int64_t>(2048, (a.nq + kWavesPerBlock - 1) / kWavesPerBlock);
    hipLaunchKernelGGL(k_finalize_multi, dim3(blocks), dim3(64 * kWavesPerBlock), (size_t)per_wave * kWavesPerBlock,
                       (hipStream_t)stream, a);
  } else {
    int threads = 128;
    int blocks = (int)((a.nq + threads - 1) / threads);
    hipLaunchKernelGGL(k_finalize, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, a);
  }
  return check(hipGetLastError());
}

}  // namespace wann
