/*
 * oracle/restate.cpp -- TEST INFRASTRUCTURE ONLY (see restate.h).
 *
 * A plain-C++ CPU restatement of the window-filtered search path of JoshEngels/RangeFilteredANN.
 * It is written from the behaviour of the reference (citations below, relative to the reference
 * checkout), not from its text: own data structures (flat arrays, one shared sorted point
 * buffer, std::thread pool), same observable results.
 *
 * Parity status: PINNED against the real reference compiled here (oracle/_ref, `make ref`) by
 * tests/test_oracle_vs_reference.py and against the golden fixtures in tests/golden/ that the
 * real reference generated (tests/golden/make_golden.py).  The reference ships no golden vectors
 * of its own for this path.
 *
 * fp32 evaluation order (the reference as compiled by g++ 11.4 -O3 -march=native, FMA host):
 *   L2   : 8 independent lanes, per 8-block acc = fma(a-b, a-b, acc); if ((d+7)&~7) % 16 == 8 the
 *          last 8-block goes first; final ((((((l0+l1)+l2)+l3)+l4)+l5)+l6)+l7   (NSGDist.h:33-69)
 *   MIPS : r = r + round(q[i]*p[i]) for the first 8*floor(d/8) elements in index order, then
 *          r = fma(q[i], p[i], r) for the tail; result -r                        (mips_point.h:60-66)
 * This file is compiled with -ffp-contract=off so every fused op below is an explicit fmaf().
 */
#include "restate.h"

#include <algorithm>
#include <atomic>
#include <climits>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <memory>
#include <mutex>
#include <optional>
#include <stdexcept>
#include <string>
#include <sys/stat.h>
#include <thread>
#include <utility>
#include <vector>

namespace {

thread_local std::string g_err;

// ------------------------------------------------------------------------------------------
// persistent thread pool with dynamic chunking (stands in for parlay::parallel_for semantics:
// independent iterations, no ordering; parlay/parallel.h:153-170)
// ------------------------------------------------------------------------------------------
class Pool {
 public:
  static Pool &get() {
    static Pool p;
    return p;
  }
  void run(int64_t n, int threads, const std::function<void(int64_t)> &f) {
    if (threads <= 0) threads = (int)std::thread::hardware_concurrency();
    if (threads < 1) threads = 1;
    if (n <= 0) return;
    if (threads == 1 || n == 1 || in_parallel_) {
      for (int64_t i = 0; i < n; i++) f(i);
      return;
    }
    std::unique_lock<std::mutex> call_lock(call_mu_);
    ensure(threads - 1);
    int64_t chunk = std::max<int64_t>(1, n / ((int64_t)threads * 16));
    {
      std::lock_guard<std::mutex> lk(mu_);
      fn_ = &f;
      n_ = n;
      chunk_ = chunk;
      next_.store(0);
      active_workers_ = threads - 1;
      pending_ = threads - 1;
      epoch_++;
    }
    cv_.notify_all();
    work();
    std::unique_lock<std::mutex> lk(mu_);
    done_cv_.wait(lk, [&] { return pending_ == 0; });
    fn_ = nullptr;
  }

 private:
  Pool() {}
  ~Pool() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
      epoch_++;
    }
    cv_.notify_all();
    for (auto &t : workers_) t.join();
  }
  void ensure(int want) {
    {
      std::lock_guard<std::mutex> lk(mu_);
      birth_epoch_ = epoch_;  // new workers must not mistake an old epoch for a posted job
    }
    while ((int)workers_.size() < want) {
      int id = (int)workers_.size();
      workers_.emplace_back([this, id] { loop(id); });
    }
  }
  void work() {
    in_parallel_ = true;
    for (;;) {
      int64_t b = next_.fetch_add(chunk_);
      if (b >= n_) break;
      int64_t e = std::min(n_, b + chunk_);
      for (int64_t i = b; i < e; i++) (*fn_)(i);
    }
    in_parallel_ = false;
  }
  void loop(int id) {
    uint64_t seen;
    {
      std::lock_guard<std::mutex> lk(mu_);
      seen = birth_epoch_;
    }
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return epoch_ != seen; });
        seen = epoch_;
        if (stop_) return;
        if (id >= active_workers_) continue;
      }
      work();
      {
        std::lock_guard<std::mutex> lk(mu_);
        pending_--;
      }
      done_cv_.notify_all();
    }
  }
  std::mutex call_mu_, mu_;
  std::condition_variable cv_, done_cv_;
  std::vector<std::thread> workers_;
  const std::function<void(int64_t)> *fn_ = nullptr;
  int64_t n_ = 0, chunk_ = 1;
  std::atomic<int64_t> next_{0};
  int active_workers_ = 0, pending_ = 0;
  uint64_t epoch_ = 0, birth_epoch_ = 0;
  bool stop_ = false;
  static thread_local bool in_parallel_;
};
thread_local bool Pool::in_parallel_ = false;

inline void parallel_for(int64_t n, int threads, const std::function<void(int64_t)> &f) {
  Pool::get().run(n, threads, f);
}

// ------------------------------------------------------------------------------------------
// numerics
// ------------------------------------------------------------------------------------------
inline uint64_t hash64_2(uint64_t x) {  // parlay/utilities.h:145-150
  x = (x ^ (x >> 30)) * UINT64_C(0xbf58476d1ce4e5b9);
  x = (x ^ (x >> 27)) * UINT64_C(0x94d049bb133111eb);
  return x ^ (x >> 31);
}

inline int hash_bits(int64_t beam) {  // beamSearch.h:66 (log2 of a long product, in double)
  double lg = std::log2((double)(beam * beam));
  return std::max<int>(10, (int)std::ceil(lg) - 2);
}

// rows are zero padded to a multiple of 8 floats, so reading up to ((d+7)&~7) is defined here
// (the reference reads uninitialised padding there: point_range.h:104-107, NSGDist.h:49).
float dist_l2(const float *a, const float *b, uint32_t d) {
  uint32_t D = (d + 7) & ~7u, DR = D % 16, DD = D - DR;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  auto block = [&](uint32_t off) {
    for (int j = 0; j < 8; j++) {
      float t = a[off + j] - b[off + j];
      acc[j] = std::fmaf(t, t, acc[j]);
    }
  };
  if (DR) block(DD);
  for (uint32_t i = 0; i < DD; i += 16) {
    block(i);
    block(i + 8);
  }
  return ((((((acc[0] + acc[1]) + acc[2]) + acc[3]) + acc[4]) + acc[5]) + acc[6]) + acc[7];
}

float dist_mips(const float *p, const float *q, uint32_t d) {
  uint32_t dv = d & ~7u;
  float r = 0.0f;
  for (uint32_t i = 0; i < dv; i++) {
    float prod = q[i] * p[i];
    r = r + prod;
  }
  for (uint32_t i = dv; i < d; i++) r = std::fmaf(q[i], p[i], r);
  return -r;
}

// uint8 / int8 point sets (metric | ORC_INTEGER): the rows hold the integer element values and the distance is the
// reference's int32 accumulation cast to float (euclidian_point.h:44-60, mips_point.h:44-58), exact for any dimension.
float dist_integer(bool mips, const float *p, const float *q, uint32_t d) {
  int32_t r = 0;
  if (mips) {
    for (uint32_t i = 0; i < d; i++) r += (int32_t)q[i] * (int32_t)p[i];
    return -((float)r);
  }
  for (uint32_t i = 0; i < d; i++) {
    const int32_t t = (int32_t)p[i] - (int32_t)q[i];
    r += t * t;
  }
  return (float)r;
}

inline float distance(int metric, const float *p, const float *q, uint32_t d) {
  if (metric & ORC_INTEGER) return dist_integer((metric & 1) == ORC_MIPS, p, q, d);
  return metric == ORC_MIPS ? dist_mips(p, q, d) : dist_l2(p, q, d);
}

// ------------------------------------------------------------------------------------------
// graph: n rows of (maxdeg+1) int32, slot 0 = degree (graph.h:115-206)
// ------------------------------------------------------------------------------------------
struct Graph {
  int64_t n = 0, maxdeg = 0;
  std::vector<int32_t> rows;
  const int32_t *row(int64_t i) const { return rows.data() + i * (maxdeg + 1); }
  int32_t *row(int64_t i) { return rows.data() + i * (maxdeg + 1); }
};

bool graph_load(const std::string &path, Graph &g) {
  FILE *f = fopen(path.c_str(), "rb");
  if (!f) return false;
  int32_t hdr[2];
  if (fread(hdr, 4, 2, f) != 2) {
    fclose(f);
    return false;
  }
  g.n = hdr[0];
  g.maxdeg = hdr[1];
  std::vector<int32_t> deg(g.n);
  if ((int64_t)fread(deg.data(), 4, g.n, f) != g.n) {
    fclose(f);
    return false;
  }
  g.rows.assign((size_t)g.n * (g.maxdeg + 1), 0);
  std::vector<int32_t> buf;
  for (int64_t i = 0; i < g.n; i++) {
    int32_t dg = deg[i];
    if (dg < 0 || dg > g.maxdeg) {
      fclose(f);
      return false;
    }
    int32_t *r = g.row(i);
    r[0] = dg;
    if (dg && (int32_t)fread(r + 1, 4, dg, f) != dg) {
      fclose(f);
      return false;
    }
  }
  fclose(f);
  return true;
}

bool graph_save(const std::string &path, const Graph &g) {
  FILE *f = fopen(path.c_str(), "wb");
  if (!f) return false;
  int32_t hdr[2] = {(int32_t)g.n, (int32_t)g.maxdeg};
  fwrite(hdr, 4, 2, f);
  std::vector<int32_t> deg(g.n);
  for (int64_t i = 0; i < g.n; i++) deg[i] = g.row(i)[0];
  fwrite(deg.data(), 4, g.n, f);
  for (int64_t i = 0; i < g.n; i++) fwrite(g.row(i) + 1, 4, deg[i], f);
  fclose(f);
  return true;
}

// ------------------------------------------------------------------------------------------
// beam search (beamSearch.h:51-184)
// ------------------------------------------------------------------------------------------
using pid = std::pair<int32_t, float>;

inline bool pid_less(const pid &a, const pid &b) {  // beamSearch.h:59-61
  return a.second < b.second || (a.second == b.second && a.first < b.first);
}

struct SearchArgs {
  const Graph *G;
  const float *pts;  // base of the shared point buffer
  int64_t stride, d;
  int metric;
  int64_t subset_start;  // Points[local] = row subset_start + local (point_range.h:179-181)
  const float *q;
  int64_t qid;  // Point::id() of the query (range_filter_tree.h:71-72; quirk H4)
  int64_t start_node, k, beam;
  double cut;
  int64_t limit, degree_limit;
};

struct SearchOut {
  std::vector<pid> beam, visited;
  int64_t dist_cmps = 0;
};

void beam_search(const SearchArgs &A, SearchOut &out) {
  const Graph &G = *A.G;
  const int64_t B = A.beam;
  const int bits = hash_bits(B);
  const uint64_t mask = (UINT64_C(1) << bits) - 1;
  std::vector<int32_t> seen((size_t)1 << bits, -1);
  auto vec = [&](int64_t local) { return A.pts + (A.subset_start + local) * A.stride; };
  auto dist_to = [&](int64_t local) { return distance(A.metric, vec(local), A.q, (uint32_t)A.d); };

  std::vector<pid> &frontier = out.beam, &visited = out.visited;
  frontier.clear();
  visited.clear();
  frontier.reserve(B);
  frontier.emplace_back((int32_t)A.start_node, dist_to(A.start_node));  // :80-82
  out.dist_cmps = 1;

  std::vector<pid> merged, cand;
  merged.reserve(B + G.maxdeg + 1);
  cand.reserve(G.maxdeg);
  std::vector<int32_t> keep;
  keep.reserve(G.maxdeg);

  bool have_next = true;
  pid next = frontier[0];
  int64_t num_visited = 0;
  const bool metric_is_metric = ((A.metric & 1) == ORC_L2);  // euclidian_point.h:71, mips_point.h:72

  while (have_next && num_visited < A.limit) {  // :108
    pid cur = next;
    visited.insert(std::upper_bound(visited.begin(), visited.end(), cur, pid_less), cur);  // :114
    num_visited++;

    const int32_t *row = G.row(cur.first);
    int64_t deg = std::min<int64_t>(row[0], A.degree_limit);  // :124
    keep.clear();
    cand.clear();
    for (int64_t i = 0; i < deg; i++) {  // :125-131
      int32_t a = row[1 + i];
      if ((int64_t)a == A.qid) continue;  // never touches the table
      uint64_t loc = hash64_2((uint64_t)(int64_t)a) & mask;
      if (seen[loc] == a) continue;
      seen[loc] = a;
      keep.push_back(a);
    }
    float cutoff = ((int64_t)frontier.size() < B) ? (float)INT_MAX : frontier.back().second;  // :135
    for (int32_t a : keep) {
      float dd = dist_to(a);
      out.dist_cmps++;
      if (dd >= cutoff) continue;
      cand.emplace_back(a, dd);
    }
    std::sort(cand.begin(), cand.end(), pid_less);  // :148
    merged.resize(frontier.size() + cand.size());
    size_t m = std::set_union(frontier.begin(), frontier.end(), cand.begin(), cand.end(),
                              merged.begin(), pid_less) -
               merged.begin();                // :151-154
    m = std::min<size_t>((size_t)B, m);       // :157
    if (A.k > 0 && (int64_t)m > A.k && metric_is_metric) {  // :162-167 (dead on the post-filter path)
      pid bound(0, (float)(A.cut * (double)merged[A.k].second));
      m = std::upper_bound(merged.begin(), merged.begin() + m, bound, pid_less) - merged.begin();
    }
    frontier.assign(merged.begin(), merged.begin() + m);

    // first frontier entry not in visited (std::set_difference, :175-178).  MULTISET semantics: an equal pair consumes ONE
    // element of each side, so a frontier that holds two copies of an entry (a row listed the node twice and the lossy filter
    // let both through) against a visited list that holds one still has a copy "unvisited" -- the reference visits it again.
    have_next = false;
    size_t vi = 0;
    for (size_t fi = 0; fi < frontier.size(); fi++) {
      while (vi < visited.size() && pid_less(visited[vi], frontier[fi])) vi++;
      if (vi < visited.size() && !pid_less(frontier[fi], visited[vi])) {  // equal: both advance
        vi++;
        continue;
      }
      next = frontier[fi];
      have_next = true;
      break;
    }
  }
}

// ------------------------------------------------------------------------------------------
// Vamana build of one partition (vamana/index.h)
// ------------------------------------------------------------------------------------------
struct BuildCtx {
  const float *pts;
  int64_t stride, d;
  int metric;
  int64_t start;  // subset_start
  int64_t n, R, L;
  double alpha;
};

inline float pdist(const BuildCtx &C, int64_t a, int64_t b) {
  return distance(C.metric, C.pts + (C.start + a) * C.stride, C.pts + (C.start + b) * C.stride,
                  (uint32_t)C.d);
}

// Order of exactly equidistant candidates.  The reference sorts by distance ONLY with std::sort (vamana/index.h:77-78,
// graph.h:106), so equal keys end up wherever libstdc++'s introsort leaves them -- a deterministic function of the
// input sequence.  The restatement does exactly that (same std::sort, same sequence, same comparator): byte-identical
// graphs also on integer-valued data, checked against files the real reference wrote.  ORC_REF_TIES=0 (read per build):
// ties break by id instead (what the product's builders do under WANN_REF_TIES=0).
inline bool ref_ties() {  // default on; ORC_REF_TIES=0: ties by id
  const char *e = getenv("ORC_REF_TIES");
  return !(e && *e == '0');
}
inline bool dist_only_less(const pid &a, const pid &b) { return a.second < b.second; }

// robustPrune (vamana/index.h:61-108).  cand carries distances to p (the visited list of the build search, sorted by
// (dist, id) like the reference's, or the incoming sources in batch order).
std::vector<int32_t> robust_prune(const BuildCtx &C, const Graph &G, int32_t p,
                                  std::vector<pid> cand, bool add = true) {
  if (add) {
    const int32_t *row = G.row(p);
    for (int32_t i = 0; i < row[0]; i++) cand.emplace_back(row[1 + i], pdist(C, row[1 + i], p));
  }
  if (ref_ties()) std::sort(cand.begin(), cand.end(), dist_only_less);
  else std::sort(cand.begin(), cand.end(), pid_less);
  std::vector<int32_t> out;
  out.reserve(C.R);
  size_t idx = 0;
  while ((int64_t)out.size() < C.R && idx < cand.size()) {
    int32_t ps = cand[idx].first;
    idx++;
    if (ps == p || ps == -1) continue;
    out.push_back(ps);
    for (size_t i = idx; i < cand.size(); i++) {
      int32_t pp = cand[i].first;
      if (pp == -1) continue;
      float d_sp = pdist(C, ps, pp);
      if (C.alpha * (double)d_sp <= (double)cand[i].second) cand[i].first = -1;  // :99
    }
  }
  return out;
}

// parlay::hash64 (parlay/utilities.h:131-141)
inline uint64_t hash64(uint64_t u) {
  uint64_t v = u * 3935559000370003845ul + 2691343689449507681ul;
  v ^= v >> 21;
  v ^= v << 37;
  v ^= v >> 4;
  v *= 4768777513237032717ul;
  v ^= v << 20;
  v ^= v >> 41;
  v ^= v << 5;
  return v;
}

// parlay::random_permutation<int>(n, random()) (parlay/random.h:79-159): below 8192 elements a Knuth
// shuffle of iota driven by r.ith_rand(i) = hash64(i + state), state 0 (:81-89,97-103); otherwise a STABLE
// counting sort of iota by hash64(i) & (2^bits - 1) (:105-125; both the sequential and the blocked
// counting sort keep equal keys in input order, internal/counting_sort.h:48-63,66-74) followed by a
// Knuth shuffle of every bucket with r.fork(bucket), state = hash64(hash64(bucket)) (:128-133, :53).
std::vector<int32_t> parlay_random_permutation(int64_t n) {
  std::vector<int32_t> out((size_t)n);
  auto knuth = [](int32_t *A, size_t len, uint64_t state) {
    if (len < 2) return;
    for (size_t i = len - 1; i > 0; i--) std::swap(A[i], A[hash64(i + state) % (i + 1)]);
  };
  if (n < 8192) {
    for (int64_t i = 0; i < n; i++) out[i] = (int32_t)i;
    knuth(out.data(), (size_t)n, 0);
    return out;
  }
  size_t lg = 0;  // log2_up (utilities.h:224-233)
  for (uint64_t b = (uint64_t)n - 1; b > 0; b >>= 1) lg++;
  const size_t bits = ((uint64_t)n < (1ull << 27)) ? (lg - 7) / 2 : (lg - 17);
  const size_t nb = (size_t)1 << bits, mask = nb - 1;
  std::vector<size_t> off(nb + 1, 0);
  for (int64_t i = 0; i < n; i++) off[(hash64((uint64_t)i) & mask) + 1]++;
  for (size_t b = 0; b < nb; b++) off[b + 1] += off[b];
  std::vector<size_t> cur(off.begin(), off.end() - 1);
  for (int64_t i = 0; i < n; i++) out[cur[hash64((uint64_t)i) & mask]++] = (int32_t)i;
  for (size_t b = 0; b < nb; b++) knuth(out.data() + off[b], off[b + 1] - off[b], hash64(hash64((uint64_t)b)));
  return out;
}

void vamana_build(const BuildCtx &C, Graph &G, int threads) {
  const int64_t n = C.n;
  G.n = n;
  G.maxdeg = C.R;
  G.rows.assign((size_t)n * (C.R + 1), 0);
  if (n == 0) return;
  // insertion order: parlay::random_permutation<int>(n) with the default generator (vamana/index.h:233)
  std::vector<int32_t> order = parlay_random_permutation(n);
  const int32_t start_point = 0;  // inserts[0] before shuffling (:128)
  size_t max_batch = std::min<size_t>((size_t)(0.02 * (double)(float)n), 1000000ul);  // :224-226
  if (max_batch == 0) max_batch = (size_t)n;
  size_t m = (size_t)n, count = 0, inc = 0;
  while (count < m) {
    size_t floor_, ceil_;
    if (std::pow(2.0, (double)inc) <= (double)max_batch) {  // :245-253
      floor_ = (size_t)std::pow(2.0, (double)inc) - 1;
      ceil_ = std::min((size_t)std::pow(2.0, (double)(inc + 1)), m) - 1;
      count = ceil_;
    } else {
      floor_ = count;
      ceil_ = std::min(count + max_batch, m);
      count += max_batch;
    }
    size_t bs = ceil_ - floor_;
    std::vector<std::vector<int32_t>> new_out(bs);
    parallel_for((int64_t)bs, threads, [&](int64_t bi) {
      int32_t index = order[floor_ + bi];
      SearchArgs A;
      A.G = &G;
      A.pts = C.pts;
      A.stride = C.stride;
      A.d = C.d;
      A.metric = C.metric;
      A.subset_start = C.start;
      A.q = C.pts + (C.start + index) * C.stride;
      A.qid = C.start + index;  // parent index: builder-side id quirk
      A.start_node = start_point;
      A.k = 0;
      A.beam = C.L;
      A.cut = 0.0;
      A.limit = n;
      A.degree_limit = G.maxdeg;  // :270
      SearchOut so;
      beam_search(A, so);
      new_out[bi] = robust_prune(C, G, index, so.visited);
    });
    // out-edges of the batch, then reverse edges grouped by target in batch order (:277-306)
    for (size_t bi = 0; bi < bs; bi++) {
      int32_t *row = G.row(order[floor_ + bi]);
      row[0] = (int32_t)new_out[bi].size();
      std::copy(new_out[bi].begin(), new_out[bi].end(), row + 1);
    }
    std::vector<std::pair<int32_t, int32_t>> rev;  // (target, source)
    for (size_t bi = 0; bi < bs; bi++)
      for (int32_t t : new_out[bi]) rev.emplace_back(t, order[floor_ + bi]);
    std::stable_sort(rev.begin(), rev.end(),
                     [](const auto &a, const auto &b) { return a.first < b.first; });
    std::vector<size_t> grp;  // group starts
    for (size_t i = 0; i < rev.size(); i++)
      if (i == 0 || rev[i].first != rev[i - 1].first) grp.push_back(i);
    grp.push_back(rev.size());
    parallel_for((int64_t)grp.size() - 1, threads, [&](int64_t gi) {
      size_t b = grp[gi], e = grp[gi + 1];
      int32_t tgt = rev[b].first;
      int32_t *row = G.row(tgt);
      if ((int64_t)(e - b) + row[0] <= C.R) {
        for (size_t i = b; i < e; i++) row[1 + row[0]++] = rev[i].second;
      } else {
        std::vector<pid> cand;
        cand.reserve(e - b);
        for (size_t i = b; i < e; i++) cand.emplace_back(rev[i].second, pdist(C, rev[i].second, tgt));
        auto no = robust_prune(C, G, tgt, std::move(cand));
        row[0] = (int32_t)no.size();
        std::copy(no.begin(), no.end(), row + 1);
      }
    });
    inc++;
  }
  // final per-node neighbour sort by distance to the node (:131-134); ties by id here
  parallel_for(n, threads, [&](int64_t i) {
    int32_t *row = G.row(i);
    std::vector<pid> nb(row[0]);
    for (int32_t j = 0; j < row[0]; j++) nb[j] = pid(row[1 + j], pdist(C, i, row[1 + j]));
    if (ref_ties()) std::sort(nb.begin(), nb.end(), dist_only_less);
    else std::sort(nb.begin(), nb.end(), pid_less);
    for (int32_t j = 0; j < row[0]; j++) row[1 + j] = nb[j].first;
  });
}

// ------------------------------------------------------------------------------------------
// index objects
// ------------------------------------------------------------------------------------------
struct Leaf {           // one partition: a contiguous slice [start, start+n) of the sorted order
  int64_t start = 0, n = 0;
  Graph G;              // vamana leaves only
  float lo = 0, hi = 0; // label range of the slice (postfilter_vamana.h:50-52)
};

struct Counters {
  std::atomic<int64_t> searches{0}, hops{0}, dist_cmps{0};
};

}  // namespace

struct orc_index {
  int kind = 0, metric = 0;
  int64_t n = 0, d = 0, stride = 0;
  std::vector<float> pts;        // padded rows; label-sorted for tree kinds, raw order otherwise
  std::vector<float> labels;     // sorted labels (tree kinds) or raw labels
  std::vector<int64_t> decoding; // sorted -> original (tree_utils.h:85)
  int32_t cutoff = 1000;
  size_t split = 2;
  float fsplit = 2.f, fshift = .5f;
  int64_t R = 64, L = 500;
  double alpha = 1.0;
  std::string cache;
  bool vamana_leaves = false;
  std::vector<std::vector<size_t>> offsets;         // WST (range_filter_tree.h:103)
  std::vector<size_t> bucket_sizes, bucket_shifts;  // super tree (:90-91)
  std::vector<std::vector<Leaf>> leaves;
  // stand-alone prefilter (prefiltering.h:33-36)
  std::vector<float> fv_sorted;
  std::vector<int32_t> fi_sorted;
  Counters ctr;
};

namespace {

std::string graph_filename(const orc_index &I, const Leaf &lf) {  // postfilter_vamana.h:126-132
  char buf[512];
  snprintf(buf, sizeof buf, "vamana_%ld_%ld_%f_%f_%f_%zu.bin", (long)I.L, (long)I.R, I.alpha,
           (double)lf.lo, (double)lf.hi, (size_t)lf.n);
  return I.cache + buf;
}

bool file_exists(const std::string &p) {
  struct stat st;
  return stat(p.c_str(), &st) == 0;
}

void make_leaf(orc_index &I, Leaf &lf, int64_t start, int64_t end, int threads) {
  lf.start = start;
  lf.n = end - start;
  if (!I.vamana_leaves) return;
  lf.lo = *std::min_element(I.labels.begin() + start, I.labels.begin() + end);
  lf.hi = *std::max_element(I.labels.begin() + start, I.labels.begin() + end);
  if (!I.cache.empty()) {
    std::string fn = graph_filename(I, lf);
    if (file_exists(fn)) {
      if (!graph_load(fn, lf.G)) throw std::runtime_error("cannot read graph file " + fn);
      if (lf.G.n != lf.n) throw std::runtime_error("graph file size mismatch " + fn);
      return;
    }
  }
  BuildCtx C{I.pts.data(), I.stride, I.d, I.metric, start, lf.n, I.R, I.L, I.alpha};
  vamana_build(C, lf.G, threads);
  if (!I.cache.empty()) graph_save(graph_filename(I, lf), lf.G);
}

// tree_utils.h:19-37
inline size_t first_ge(float v, const std::vector<float> &fv) {
  if (fv[0] >= v) return 0;
  size_t s = 0, e = fv.size();
  while (s + 1 < e) {
    size_t mid = (s + e) / 2;
    if (fv[mid] >= v) e = mid;
    else s = mid;
  }
  return e;
}

struct QueryCtx {
  orc_index *I;
  const float *q;  // padded query
  int64_t qid;
  orc_qparams qp;
};

// postfilter_vamana.h:223-254
std::vector<pid> raw_query(QueryCtx &Q, const Leaf &lf, float lo, float hi, int64_t beam, bool map_ids) {
  orc_index &I = *Q.I;
  SearchArgs A;
  A.G = &lf.G;
  A.pts = I.pts.data();
  A.stride = I.stride;
  A.d = I.d;
  A.metric = I.metric;
  A.subset_start = lf.start;
  A.q = Q.q;
  A.qid = Q.qid;
  A.start_node = 0;
  A.k = beam;  // postfilter_vamana.h:145-146,169-170,178-179
  A.beam = beam;
  A.cut = Q.qp.cut;
  A.limit = Q.qp.limit;
  A.degree_limit = Q.qp.degree_limit;
  SearchOut so;
  beam_search(A, so);
  I.ctr.searches++;
  I.ctr.hops += (int64_t)so.visited.size();
  I.ctr.dist_cmps += so.dist_cmps;
  std::vector<pid> out;
  for (auto &p : so.beam) {
    float fv = I.labels[lf.start + p.first];
    if (fv >= lo && fv <= hi) out.emplace_back(map_ids ? (int32_t)(lf.start + p.first) : p.first, p.second);
  }
  return out;
}

// postfilter_vamana.h:141-188
std::vector<pid> postfilter_query(QueryCtx &Q, const Leaf &lf, float lo, float hi, const orc_qparams &qp,
                                  bool map_ids) {
  size_t knn = (size_t)qp.k;
  int64_t beam = qp.beam;
  std::vector<pid> F;
  while (F.size() < knn && beam < qp.max_beam) {
    F = raw_query(Q, lf, lo, hi, beam, map_ids);
    if (F.size() < knn) beam *= 2;
  }
  size_t fb = std::min<size_t>((size_t)(beam * qp.final_beam_multiply), (size_t)qp.max_beam);
  if (fb > (size_t)beam) F = raw_query(Q, lf, lo, hi, (int64_t)fb, map_ids);
  return F;
}

// prefiltering.h:154-204 on a slice [start, start+n) of the sorted order (subset leaves) --
// the slice of sorted labels is itself sorted, so filter_indices_sorted is the identity there.
std::vector<pid> prefilter_leaf_query(QueryCtx &Q, int64_t start, int64_t n, float lo, float hi, size_t knn) {
  orc_index &I = *Q.I;
  const float *fv = I.labels.data() + start;
  auto bs = [&](float v) {
    size_t l = 0, r = (size_t)n - 1;  // r = n-1: the last point is never included (quirk #8)
    while (l < r) {
      size_t mid = (l + r) / 2;
      if (fv[mid] < v) l = mid + 1;
      else r = mid;
    }
    return l;
  };
  size_t s = bs(lo), e = bs(hi);
  std::vector<pid> F;
  for (size_t j = s; j < e; j++)
    F.emplace_back((int32_t)(start + j),
                   distance(I.metric, I.pts.data() + (start + j) * I.stride, Q.q, (uint32_t)I.d));
  I.ctr.dist_cmps += (int64_t)(e > s ? e - s : 0);
  std::stable_sort(F.begin(), F.end(), pid_less);  // reference: unstable, by distance only
  if (F.size() > knn) F.resize(knn);
  return F;
}

std::vector<pid> leaf_query(QueryCtx &Q, const Leaf &lf, float lo, float hi, const orc_qparams &qp) {
  if (Q.I->vamana_leaves) return postfilter_query(Q, lf, lo, hi, qp, true);
  return prefilter_leaf_query(Q, lf.start, lf.n, lo, hi, (size_t)qp.k);
}

bool check_empty(const orc_index &I, float lo, float hi) {  // range_filter_tree.h:191-203
  return hi < I.labels.front() || lo > I.labels.back();
}

void sort_and_truncate(std::vector<pid> &F, size_t k) {  // :542-549 (by distance; ties by id here)
  std::stable_sort(F.begin(), F.end(), pid_less);
  if (F.size() > k) F.resize(k);
}

struct SeqBuckets {
  size_t row, first, last, cover_start, cover_end;
};

size_t find_range_containing(const orc_index &I, size_t row, size_t index) {  // :213-232
  const auto &off = I.offsets[row];
  size_t left = 0, right = off.size() - 1;
  while (left < right) {
    size_t mid = (left + right) / 2;
    if (index >= off[mid] && index < off[mid + 1]) return mid;
    else if (index < off[mid]) right = mid;
    else left = mid;
  }
  throw std::runtime_error("This should not be possible if index is within the filter range");
}

std::optional<SeqBuckets> largest_ranges_within(const orc_index &I, size_t istart, size_t eend) {  // :234-295
  size_t range_size = eend - istart;
  std::optional<size_t> first_row;
  for (size_t row = 0; row < I.offsets.size(); row++) {
    size_t bsz = I.offsets[row][1] - I.offsets[row][0] - 1;
    if (bsz <= range_size) {
      first_row = row;
      break;
    }
  }
  if (!first_row) return std::nullopt;
  size_t row = *first_row;
  size_t first = istart == 0 ? 0 : find_range_containing(I, row, istart - 1) + 1;
  size_t start = I.offsets.at(row).at(first);
  size_t end = I.offsets.at(row).at(first + 1);
  if (end > eend) {
    row += 1;
    if (row >= I.offsets.size()) return std::nullopt;
    first = istart == 0 ? 0 : find_range_containing(I, row, istart - 1) + 1;
    start = I.offsets.at(row).at(first);
    end = I.offsets.at(row).at(first + 1);
  }
  size_t last = first + 1;
  while (last < I.offsets[row].size() - 1) {
    size_t next_end = I.offsets[row].at(last + 1);
    if (next_end > eend) break;
    last++;
    end = next_end;
  }
  return SeqBuckets{row, first, last, start, end};
}

std::vector<pid> brute_range(QueryCtx &Q, size_t a, size_t b, std::vector<pid> &F) {
  orc_index &I = *Q.I;
  for (size_t i = a; i < b; i++)
    F.emplace_back((int32_t)i, distance(I.metric, I.pts.data() + i * I.stride, Q.q, (uint32_t)I.d));
  if (b > a) I.ctr.dist_cmps += (int64_t)(b - a);
  return F;
}

std::vector<pid> fenwick_search(QueryCtx &Q, float lo, float hi, const orc_qparams &qp) {  // :297-401
  orc_index &I = *Q.I;
  if (check_empty(I, lo, hi)) return {};
  size_t knn = (size_t)qp.k;
  size_t istart = first_ge(lo, I.labels), eend = first_ge(hi, I.labels);
  auto centre = largest_ranges_within(I, istart, eend);
  std::vector<std::pair<size_t, size_t>> to_search;
  std::optional<size_t> cov_s, cov_e;
  if (centre) {
    for (size_t b = centre->first; b < centre->last; b++) to_search.emplace_back(centre->row, b);
    cov_s = centre->cover_start;
    cov_e = centre->cover_end;
    size_t left = centre->first, right = centre->last - 1;
    for (size_t row = centre->row + 1; row < I.offsets.size(); row++) {
      left *= I.split;
      right = right * I.split + I.split - 1;
      while (left > 0) {
        size_t nls = I.offsets.at(row).at(left - 1);
        if (nls < istart) break;
        cov_s = nls;
        left -= 1;
        to_search.emplace_back(row, left);
      }
      while (right < I.offsets[row].size() - 2) {
        size_t nre = I.offsets.at(row).at(right + 2);
        if (nre > eend) break;
        cov_e = nre;
        right += 1;
        to_search.emplace_back(row, right);
      }
    }
  }
  std::vector<pid> F;
  for (auto &pr : to_search) {
    auto r = leaf_query(Q, I.leaves.at(pr.first).at(pr.second), lo, hi, qp);
    F.insert(F.end(), r.begin(), r.end());
  }
  if (cov_s && cov_e) {
    brute_range(Q, istart, *cov_s, F);
    brute_range(Q, *cov_e, eend, F);
  } else {
    brute_range(Q, istart, eend, F);
  }
  sort_and_truncate(F, knn);
  return F;
}

std::vector<pid> optimized_postfilter_search(QueryCtx &Q, float lo, float hi, const orc_qparams &qp) {  // :403-471
  orc_index &I = *Q.I;
  if (check_empty(I, lo, hi)) return {};
  size_t istart = first_ge(lo, I.labels), eend = first_ge(hi, I.labels);
  // NB: size_t arithmetic as in the reference (eend < istart wraps; 4*w compared as size_t vs int)
  if (4 * (eend - istart) < (size_t)(int64_t)I.cutoff) return fenwick_search(Q, lo, hi, qp);
  size_t row = 0, idx = 0;
  while (row + 1 < I.offsets.size()) {
    size_t nrow = row + 1;
    std::optional<size_t> nidx;
    for (size_t c = idx * I.split; c < idx * I.split + I.split; c++) {
      if (c >= I.leaves.at(nrow).size()) break;
      size_t ns = I.offsets[nrow][c], ne = I.offsets[nrow][c + 1];
      if (istart >= ns && eend <= ne) nidx = c;
    }
    if (!nidx) break;
    idx = *nidx;
    row = nrow;
  }
  size_t bsz = I.offsets[row][idx + 1] - I.offsets[row][idx];
  float ratio = (float)bsz / (eend - istart);
  if (qp.has_ratio && ratio > qp.ratio) return fenwick_search(Q, lo, hi, qp);
  return leaf_query(Q, I.leaves[row][idx], lo, hi, qp);
}

std::vector<pid> three_split_search(QueryCtx &Q, float lo, float hi, const orc_qparams &qp) {  // :473-540
  orc_index &I = *Q.I;
  if (check_empty(I, lo, hi)) return {};
  size_t istart = first_ge(lo, I.labels), eend = first_ge(hi, I.labels);
  auto centre = largest_ranges_within(I, istart, eend);
  orc_qparams qf = qp;
  qf.final_beam_multiply = 1;
  if (!centre) return fenwick_search(Q, lo, hi, qf);
  std::vector<pid> F;
  for (size_t b = centre->first; b < centre->last; b++) {
    auto r = leaf_query(Q, I.leaves.at(centre->row).at(b), lo, hi, qf);
    F.insert(F.end(), r.begin(), r.end());
  }
  size_t left_space = centre->cover_start - istart, right_space = eend - centre->cover_end;
  if (left_space > 0) {
    auto r = optimized_postfilter_search(Q, lo, I.labels[centre->cover_start], qp);
    F.insert(F.end(), r.begin(), r.end());
  }
  if (right_space > 0) {
    auto r = optimized_postfilter_search(Q, I.labels[centre->cover_end], hi, qp);
    F.insert(F.end(), r.begin(), r.end());
  }
  sort_and_truncate(F, (size_t)qp.k);
  return F;
}

std::vector<pid> super_search(QueryCtx &Q, float lo, float hi, const orc_qparams &qp) {  // super_...h:187-270
  orc_index &I = *Q.I;
  if (check_empty(I, lo, hi)) return {};
  size_t istart = first_ge(lo, I.labels), eend = first_ge(hi, I.labels);
  int64_t row, idx = 0;
  for (row = (int64_t)I.bucket_sizes.size() - 1; row >= 0; row--) {
    if (row == 0) {
      idx = 0;
      break;
    }
    size_t bsz = I.bucket_sizes[row];
    if (bsz < eend - istart) continue;
    size_t shift = I.bucket_shifts[row];
    size_t fp = istart / shift, lp = (eend - 1) / shift;
    fp = std::min(fp, I.leaves[row].size() - 1);
    lp = std::min(lp, I.leaves[row].size() - 1);
    bool found = false;
    for (size_t t = fp; t <= lp; t++) {
      size_t bs = t * shift, be = std::min(bs + bsz, (size_t)I.n);
      if (istart >= bs && eend <= be) {
        idx = (int64_t)t;
        found = true;
        break;
      }
    }
    if (found) break;
  }
  return leaf_query(Q, I.leaves[row][idx], lo, hi, qp);
}

void build_wst(orc_index &I, int threads) {  // range_filter_tree.h:129-189
  size_t n = (size_t)I.n;
  I.offsets.push_back({0, n});
  I.leaves.emplace_back(1);
  make_leaf(I, I.leaves[0][0], 0, (int64_t)n, threads);
  while (I.offsets.back().at(1) > (size_t)(int64_t)I.cutoff) {
    const auto &prev = I.offsets.back();
    size_t last_nb = prev.size() - 1;
    std::vector<size_t> off(last_nb * I.split + 1);
    off.back() = n;
    for (size_t b = 0; b < last_nb; b++) {
      size_t ls = prev[b], le = prev[b + 1], lsz = le - ls;
      size_t large = (lsz + I.split - 1) / I.split, small = large - 1;
      size_t n_large = lsz - small * I.split;
      for (size_t i = 0; i < I.split; i++)
        off[b * I.split + i] = i < n_large ? ls + i * large : ls + n_large * large + (i - n_large) * small;
    }
    I.offsets.push_back(off);
    I.leaves.emplace_back(last_nb * I.split);
    auto &lv = I.leaves.back();
    // big partitions: parallel inside the build; small ones: parallel across partitions
    bool inner = (n / lv.size()) >= 20000;
    if (inner) {
      for (size_t b = 0; b < lv.size(); b++) make_leaf(I, lv[b], (int64_t)off[b], (int64_t)off[b + 1], threads);
    } else {
      parallel_for((int64_t)lv.size(), threads,
                   [&](int64_t b) { make_leaf(I, lv[b], (int64_t)off[b], (int64_t)off[b + 1], 1); });
    }
  }
}

void build_super(orc_index &I, int threads) {  // super_optimized_postfilter_tree.h:118-171
  if (I.fsplit <= 1) throw std::runtime_error("split_factor must be greater than 1");
  if (I.fshift >= 1 || I.fshift <= 0) throw std::runtime_error("shift_factor must be between 0 and 1");
  size_t n = (size_t)I.n;
  I.leaves.emplace_back(1);
  make_leaf(I, I.leaves[0][0], 0, (int64_t)n, threads);
  I.bucket_sizes.push_back(n);
  I.bucket_shifts.push_back(0);
  while (I.bucket_sizes.back() > (size_t)(int64_t)I.cutoff) {
    size_t last = I.bucket_sizes.back();
    size_t bsz = (size_t)(((float)last + I.fsplit - 1) / I.fsplit);  // float arithmetic (:148-149)
    size_t shift = (size_t)std::ceil((float)bsz * I.fshift);          // :150
    I.bucket_sizes.push_back(bsz);
    I.bucket_shifts.push_back(shift);
    size_t nb = ((n - bsz) + shift - 1) / shift + 1;  // :159-160
    I.leaves.emplace_back(nb);
    auto &lv = I.leaves.back();
    bool inner = bsz >= 20000;
    auto mk = [&](size_t b, int t) {
      size_t bs = b * shift, be = std::min(bs + bsz, n);
      make_leaf(I, lv[b], (int64_t)bs, (int64_t)be, t);
    };
    if (inner) for (size_t b = 0; b < nb; b++) mk(b, threads);
    else parallel_for((int64_t)nb, threads, [&](int64_t b) { mk((size_t)b, 1); });
  }
}

}  // namespace

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
extern "C" {

uint64_t orc_hash64_2(uint64_t x) { return hash64_2(x); }
void orc_random_permutation(int64_t n, int32_t *out) {
  std::vector<int32_t> p = parlay_random_permutation(n);
  memcpy(out, p.data(), (size_t)n * 4);
}
int orc_hash_bits(int64_t beam) { return hash_bits(beam); }
float orc_distance(int metric, const float *p, const float *q, uint32_t d) {
  // defensive copy into zero padded buffers so callers may pass unpadded rows
  uint32_t D = (d + 7) & ~7u;
  std::vector<float> a(D, 0.f), b(D, 0.f);
  memcpy(a.data(), p, d * 4);
  memcpy(b.data(), q, d * 4);
  return distance(metric, a.data(), b.data(), d);
}
const char *orc_last_error(void) { return g_err.c_str(); }
void orc_free(void *p) { free(p); }

int64_t orc_beam_search(const int32_t *graph, int64_t n, int64_t maxdeg, const float *points,
                        int64_t stride, int64_t d, int metric, int64_t subset_start,
                        const float *query, int64_t query_id, int64_t start_node, int64_t k,
                        int64_t beam, double cut, int64_t limit, int64_t degree_limit,
                        int32_t *out_ids, float *out_dists, int32_t *vis_ids, float *vis_dists,
                        int64_t *n_visited, int64_t *dist_cmps) {
  Graph G;
  G.n = n;
  G.maxdeg = maxdeg;
  G.rows.assign(graph, graph + (size_t)n * (maxdeg + 1));
  std::vector<float> q(((size_t)d + 7) & ~(size_t)7, 0.f);
  memcpy(q.data(), query, d * 4);
  SearchArgs A{&G, points, stride, d, metric, subset_start, q.data(), query_id, start_node, k, beam, cut, limit, degree_limit};
  SearchOut so;
  beam_search(A, so);
  for (size_t i = 0; i < so.beam.size(); i++) {
    out_ids[i] = so.beam[i].first;
    out_dists[i] = so.beam[i].second;
  }
  if (vis_ids)
    for (size_t i = 0; i < so.visited.size(); i++) {
      vis_ids[i] = so.visited[i].first;
      vis_dists[i] = so.visited[i].second;
    }
  if (n_visited) *n_visited = (int64_t)so.visited.size();
  if (dist_cmps) *dist_cmps = so.dist_cmps;
  return (int64_t)so.beam.size();
}

int orc_graph_load(const char *path, int32_t **rows, int64_t *n, int64_t *maxdeg) {
  Graph g;
  if (!graph_load(path, g)) return 1;
  *n = g.n;
  *maxdeg = g.maxdeg;
  *rows = (int32_t *)malloc(g.rows.size() * 4 + 4);
  memcpy(*rows, g.rows.data(), g.rows.size() * 4);
  return 0;
}

int orc_graph_save(const char *path, const int32_t *rows, int64_t n, int64_t maxdeg) {
  Graph g;
  g.n = n;
  g.maxdeg = maxdeg;
  g.rows.assign(rows, rows + (size_t)n * (maxdeg + 1));
  return graph_save(path, g) ? 0 : 1;
}

int orc_vamana_build(const float *points, int64_t stride, int64_t d, int metric,
                     int64_t subset_start, int64_t n, int64_t R, int64_t L, double alpha,
                     int32_t *rows, int threads) {
  try {
    BuildCtx C{points, stride, d, metric, subset_start, n, R, L, alpha};
    Graph G;
    vamana_build(C, G, threads);
    memcpy(rows, G.rows.data(), G.rows.size() * 4);
    return 0;
  } catch (std::exception &e) {
    g_err = e.what();
    return 1;
  }
}

orc_index *orc_index_create(int kind, int metric, const float *points, int64_t n, int64_t d,
                            const float *labels, int32_t cutoff, double split_factor,
                            double shift_factor, int64_t R, int64_t L, double alpha,
                            const char *cache_path, int threads) {
  std::unique_ptr<orc_index> I(new orc_index);
  try {
    I->kind = kind;
    I->metric = metric;
    I->n = n;
    I->d = d;
    I->stride = ((d * 4 + 63) / 64) * 64 / 4;  // point_range.h:39-44 (64-byte rows), zero padded
    I->cutoff = cutoff;
    I->split = (size_t)split_factor;
    I->fsplit = (float)split_factor;
    I->fshift = (float)shift_factor;
    I->R = R;
    I->L = L;
    I->alpha = alpha;
    I->cache = cache_path ? cache_path : "";
    I->vamana_leaves = (kind == ORC_POSTFILTER || kind == ORC_TREE_VAMANA || kind == ORC_SUPER);
    if (n <= 0 || d <= 0) throw std::runtime_error("empty point set");
    bool sorted_kinds = (kind == ORC_TREE_PREFILTER || kind == ORC_TREE_VAMANA || kind == ORC_SUPER);
    std::vector<int64_t> order(n);
    for (int64_t i = 0; i < n; i++) order[i] = i;
    if (sorted_kinds)  // tree_utils.h:68-73 (unstable there; canonical = stable by (label, id))
      std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return labels[a] < labels[b]; });
    I->pts.assign((size_t)n * I->stride, 0.f);
    I->labels.resize(n);
    I->decoding = order;
    for (int64_t s = 0; s < n; s++) {
      memcpy(I->pts.data() + s * I->stride, points + order[s] * d, d * 4);
      I->labels[s] = labels[order[s]];
    }
    switch (kind) {
      case ORC_PREFILTER: {  // prefiltering.h:76-122
        I->fi_sorted.resize(n);
        for (int64_t i = 0; i < n; i++) I->fi_sorted[i] = (int32_t)i;
        std::stable_sort(I->fi_sorted.begin(), I->fi_sorted.end(),
                         [&](int32_t a, int32_t b) { return labels[a] < labels[b]; });
        I->fv_sorted.resize(n);
        for (int64_t i = 0; i < n; i++) I->fv_sorted[i] = labels[I->fi_sorted[i]];
        break;
      }
      case ORC_POSTFILTER: {
        I->leaves.emplace_back(1);
        make_leaf(*I, I->leaves[0][0], 0, n, threads);
        break;
      }
      case ORC_TREE_PREFILTER:
      case ORC_TREE_VAMANA:
        if (I->split < 2) throw std::runtime_error("split_factor must be >= 2");
        build_wst(*I, threads);
        break;
      case ORC_SUPER:
        build_super(*I, threads);
        break;
      default:
        throw std::runtime_error("unknown index kind");
    }
  } catch (std::exception &e) {
    g_err = e.what();
    return nullptr;
  }
  return I.release();
}

void orc_index_destroy(orc_index *I) { delete I; }

int orc_batch_search(orc_index *I, const float *queries, const float *ranges, int64_t nq,
                     const char *method, const orc_qparams *qp, uint32_t *ids, float *dists,
                     int threads, int64_t *counters) {
  std::string m = method ? method : "";
  I->ctr.searches = 0;
  I->ctr.hops = 0;
  I->ctr.dist_cmps = 0;
  std::atomic<bool> failed{false};
  std::string err;
  std::mutex err_mu;
  const int64_t k = qp->k;
  parallel_for(nq, threads, [&](int64_t i) {
    try {
      std::vector<float> q((size_t)I->stride, 0.f);
      memcpy(q.data(), queries + i * I->d, I->d * 4);
      QueryCtx Q{I, q.data(), i, *qp};
      float lo = ranges[2 * i], hi = ranges[2 * i + 1];
      std::vector<pid> res;
      uint32_t pad_id = 0;
      bool decode = true;
      switch (I->kind) {
        case ORC_TREE_PREFILTER:
        case ORC_TREE_VAMANA:
          if (m == "optimized_postfilter") res = optimized_postfilter_search(Q, lo, hi, *qp);
          else if (m == "three_split") res = three_split_search(Q, lo, hi, *qp);
          else res = fenwick_search(Q, lo, hi, *qp);
          break;
        case ORC_SUPER:
          res = super_search(Q, lo, hi, *qp);
          break;
        case ORC_POSTFILTER:  // postfilter_vamana.h:191-219: raw ids, pad id -1
          res = postfilter_query(Q, I->leaves[0][0], lo, hi, *qp, false);
          pad_id = 0xFFFFFFFFu;
          decode = false;
          break;
        case ORC_PREFILTER: {  // prefiltering.h:124-146,154-204 (global arrays, gather via argsort)
          auto bs = [&](float v) {
            size_t l = 0, r = (size_t)I->n - 1;
            while (l < r) {
              size_t mid = (l + r) / 2;
              if (I->fv_sorted[mid] < v) l = mid + 1;
              else r = mid;
            }
            return l;
          };
          size_t s = bs(lo), e = bs(hi);
          for (size_t j = s; j < e; j++) {
            int32_t idx = I->fi_sorted[j];
            res.emplace_back(idx, distance(I->metric, I->pts.data() + (int64_t)idx * I->stride, q.data(), (uint32_t)I->d));
          }
          if (e > s) I->ctr.dist_cmps += (int64_t)(e - s);
          std::stable_sort(res.begin(), res.end(), pid_less);
          if ((int64_t)res.size() > k) res.resize(k);
          pad_id = 0xFFFFFFFFu;  // reference reads past the end here (UB); defined padding instead
          decode = false;
          break;
        }
      }
      for (int64_t j = 0; j < k; j++) {
        if (j < (int64_t)res.size()) {
          ids[i * k + j] = decode ? (uint32_t)I->decoding.at(res[j].first) : (uint32_t)res[j].first;
          dists[i * k + j] = res[j].second;
        } else {
          ids[i * k + j] = pad_id;
          dists[i * k + j] = std::numeric_limits<float>::max();
        }
      }
    } catch (std::exception &e) {
      std::lock_guard<std::mutex> lk(err_mu);
      failed = true;
      err = e.what();
    }
  });
  if (counters) {
    counters[0] = I->ctr.searches;
    counters[1] = I->ctr.hops;
    counters[2] = I->ctr.dist_cmps;
  }
  if (failed) {
    g_err = err;
    return 1;
  }
  return 0;
}

int64_t orc_num_levels(const orc_index *I) { return (int64_t)I->leaves.size(); }
int64_t orc_level_size(const orc_index *I, int64_t level) { return (int64_t)I->leaves.at(level).size(); }
int orc_partition_range(const orc_index *I, int64_t level, int64_t idx, int64_t *start, int64_t *end) {
  const Leaf &lf = I->leaves.at(level).at(idx);
  *start = lf.start;
  *end = lf.start + lf.n;
  return 0;
}
const int32_t *orc_partition_graph(const orc_index *I, int64_t level, int64_t idx, int64_t *n, int64_t *maxdeg) {
  const Leaf &lf = I->leaves.at(level).at(idx);
  *n = lf.G.n;
  *maxdeg = lf.G.maxdeg;
  return lf.G.rows.data();
}
const int64_t *orc_decoding(const orc_index *I) { return I->decoding.data(); }
const float *orc_sorted_labels(const orc_index *I) { return I->labels.data(); }

}  // extern "C"
