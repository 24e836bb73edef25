// wann_build.cpp -- host-side index construction (see wann_build.h for the reference citations).
//
// The Vamana builder follows the reference's algorithm (prefix-doubling batches over a fixed
// insertion order, beam search on the snapshot, robustPrune, reverse edges appended or re-pruned,
// final per-node neighbour sort) with this engine's own data structures: the search frontier is
// one sorted array of 64-bit keys (order-preserving distance bits | node id | visited bit) merged
// in place, which is also the representation the gfx950 search kernel uses.  Insertion order is
// argsort(hash64_2(i)) and distance ties break by id, so builds are deterministic for any thread
// count (they are not byte-identical to the reference builder's graphs; graphs written by the
// reference load through the same cache files).
#include "wann_build.h"

#include <algorithm>
#include <atomic>
#include <climits>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>

namespace wann {

// ------------------------------------------------------------------------------------------------
// thread pool
// ------------------------------------------------------------------------------------------------
namespace {
class WorkerPool {
 public:
  static WorkerPool &instance() {
    static WorkerPool p;
    return p;
  }
  void run(int64_t n, int threads, const std::function<void(int64_t)> &f) {
    if (n <= 0) return;
    if (threads <= 1 || n == 1 || nested_) {
      for (int64_t i = 0; i < n; i++) f(i);
      return;
    }
    std::lock_guard<std::mutex> serial(entry_);
    grow(threads - 1);
    {
      std::lock_guard<std::mutex> lk(m_);
      job_ = &f;
      total_ = n;
      grain_ = std::max<int64_t>(1, n / ((int64_t)threads * 8));
      cursor_.store(0, std::memory_order_relaxed);
      helpers_ = threads - 1;
      outstanding_ = threads - 1;
      generation_++;
    }
    wake_.notify_all();
    drain();
    std::unique_lock<std::mutex> lk(m_);
    idle_.wait(lk, [&] { return outstanding_ == 0; });
    job_ = nullptr;
  }

 private:
  ~WorkerPool() {
    {
      std::lock_guard<std::mutex> lk(m_);
      quit_ = true;
      generation_++;
    }
    wake_.notify_all();
    for (auto &t : threads_) t.join();
  }
  void grow(int want) {
    {
      std::lock_guard<std::mutex> lk(m_);
      birth_generation_ = generation_;  // a new worker must not take an old generation for a job
    }
    while ((int)threads_.size() < want) {
      int id = (int)threads_.size();
      threads_.emplace_back([this, id] { worker(id); });
    }
  }
  void drain() {
    nested_ = true;
    for (;;) {
      int64_t b = cursor_.fetch_add(grain_, std::memory_order_relaxed);
      if (b >= total_) break;
      int64_t e = std::min(total_, b + grain_);
      for (int64_t i = b; i < e; i++) (*job_)(i);
    }
    nested_ = false;
  }
  void worker(int id) {
    uint64_t seen;
    {
      std::lock_guard<std::mutex> lk(m_);
      seen = birth_generation_;
    }
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(m_);
        wake_.wait(lk, [&] { return generation_ != seen; });
        seen = generation_;
        if (quit_) return;
        if (id >= helpers_) continue;
      }
      drain();
      {
        std::lock_guard<std::mutex> lk(m_);
        outstanding_--;
      }
      idle_.notify_all();
    }
  }
  std::mutex entry_, m_;
  std::condition_variable wake_, idle_;
  std::vector<std::thread> threads_;
  const std::function<void(int64_t)> *job_ = nullptr;
  int64_t total_ = 0, grain_ = 1;
  std::atomic<int64_t> cursor_{0};
  int helpers_ = 0, outstanding_ = 0;
  uint64_t generation_ = 0, birth_generation_ = 0;
  bool quit_ = false;
  static thread_local bool nested_;
};
thread_local bool WorkerPool::nested_ = false;
}  // namespace

void parallel_for(int64_t n, int threads, const std::function<void(int64_t)> &f) {
  WorkerPool::instance().run(n, threads, f);
}

int default_threads() {
  const char *env = getenv("PARLAY_NUM_THREADS");  // the reference's knob (run_our_method.py:131)
  if (env && atoi(env) > 0) return atoi(env);
  int hc = (int)std::thread::hardware_concurrency();
  return hc > 0 ? hc : 1;
}

// ------------------------------------------------------------------------------------------------
// numerics (same evaluation order as the kernels; built with -ffp-contract=off)
// ------------------------------------------------------------------------------------------------
static inline uint64_t mix64(uint64_t x) {  // parlay::hash64_2
  x = (x ^ (x >> 30)) * UINT64_C(0xbf58476d1ce4e5b9);
  x = (x ^ (x >> 27)) * UINT64_C(0x94d049bb133111eb);
  return x ^ (x >> 31);
}

static inline float l2_ref_order(const float *a, const float *b, int d) {
  const int D8 = (d + 7) >> 3;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const bool odd = D8 & 1;
  for (int i = 0; i < D8; i++) {
    const int blk = odd ? (i == 0 ? D8 - 1 : i - 1) : i;
    const float *pa = a + 8 * blk, *pb = b + 8 * blk;
    for (int j = 0; j < 8; j++) {
      float t = pa[j] - pb[j];
      acc[j] = std::fmaf(t, t, acc[j]);
    }
  }
  return ((((((acc[0] + acc[1]) + acc[2]) + acc[3]) + acc[4]) + acc[5]) + acc[6]) + acc[7];
}

static inline float mips_ref_order(const float *p, const float *q, int d) {
  const int dv = d & ~7;
  float r = 0.f;
  for (int i = 0; i < dv; i++) {
    volatile float prod = q[i] * p[i];  // keep the product rounded on its own
    r = r + prod;
  }
  for (int i = dv; i < d; i++) r = std::fmaf(q[i], p[i], r);
  return -r;
}

float host_distance(int metric, const float *p, const float *q, int d) {
  return metric == 1 ? mips_ref_order(p, q, d) : l2_ref_order(p, q, d);
}

static inline uint32_t fkey(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
static inline float funkey(uint32_t k) {
  uint32_t u = (k & 0x80000000u) ? (k ^ 0x80000000u) : ~k;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
static inline uint64_t mkkey(float dist, int32_t id) { return ((uint64_t)fkey(dist) << 32) | ((uint64_t)(uint32_t)id << 1); }
static inline int32_t key_id(uint64_t k) { return (int32_t)((uint32_t)k >> 1); }
static inline float key_dist(uint64_t k) { return funkey((uint32_t)(k >> 32)); }

// ------------------------------------------------------------------------------------------------
// graph cache files: [n:i32][maxDeg:i32][deg[n]:i32][edges:i32 ...]
// ------------------------------------------------------------------------------------------------
bool graph_file_load(const std::string &path, HostGraph &g) {
  FILE *f = fopen(path.c_str(), "rb");
  if (!f) return false;
  bool ok = false;
  do {
    int32_t head[2];
    if (fread(head, 4, 2, f) != 2 || head[0] < 0 || head[1] < 0) break;
    g.n = head[0];
    g.maxdeg = head[1];
    std::vector<int32_t> deg((size_t)g.n);
    if (g.n && fread(deg.data(), 4, (size_t)g.n, f) != (size_t)g.n) break;
    int64_t total = 0;
    bool bad = false;
    for (int32_t x : deg) {
      if (x < 0 || x > g.maxdeg) bad = true;
      total += x;
    }
    if (bad) break;
    std::vector<int32_t> edges((size_t)total);
    if (total && fread(edges.data(), 4, (size_t)total, f) != (size_t)total) break;
    g.rows.assign((size_t)g.n * (g.maxdeg + 1), 0);
    size_t at = 0;
    for (int64_t i = 0; i < g.n; i++) {
      int32_t *r = g.row(i);
      r[0] = deg[i];
      memcpy(r + 1, edges.data() + at, (size_t)deg[i] * 4);
      at += deg[i];
    }
    ok = true;
  } while (0);
  fclose(f);
  return ok;
}

bool graph_file_save(const std::string &path, const HostGraph &g) {
  std::string tmp = path + ".tmp" + std::to_string((long)getpid());
  FILE *f = fopen(tmp.c_str(), "wb");
  if (!f) return false;
  int32_t head[2] = {(int32_t)g.n, g.maxdeg};
  std::vector<int32_t> deg((size_t)g.n), edges;
  for (int64_t i = 0; i < g.n; i++) {
    const int32_t *r = g.row(i);
    deg[i] = r[0];
    edges.insert(edges.end(), r + 1, r + 1 + r[0]);
  }
  bool ok = fwrite(head, 4, 2, f) == 2 && fwrite(deg.data(), 4, deg.size(), f) == deg.size() &&
            fwrite(edges.data(), 4, edges.size(), f) == edges.size();
  ok = (fclose(f) == 0) && ok;
  if (ok) ok = rename(tmp.c_str(), path.c_str()) == 0;  // atomic publish: ranks share cache dirs
  if (!ok) remove(tmp.c_str());
  return ok;
}

std::string graph_file_name(const BuildSpec &s, float lo, float hi, int64_t n) {
  char buf[400];  // std::to_string(double/float) == "%f"
  snprintf(buf, sizeof buf, "vamana_%ld_%ld_%f_%f_%f_%zu.bin", (long)s.L, (long)s.R, s.alpha, (double)lo,
           (double)hi, (size_t)n);
  return s.cache + buf;
}

static bool exists(const std::string &p) {
  struct stat st;
  return stat(p.c_str(), &st) == 0;
}

// ------------------------------------------------------------------------------------------------
// Vamana build
// ------------------------------------------------------------------------------------------------
namespace {

struct BuildView {
  const float *pts;
  int64_t stride, d;
  int metric;
  int64_t start, n, R, L;
  double alpha;
  const float *vec(int64_t local) const { return pts + (start + local) * stride; }
  float dist(int64_t a, int64_t b) const { return host_distance(metric, vec(a), vec(b), (int)d); }
};

// per-thread scratch of the build-time search
struct Scratch {
  std::vector<int32_t> table;
  std::vector<uint64_t> front, cand, keep, visited;
};

// Beam search of the build (vamana/index.h:270: k = 0, beam = L, limit = n, degree_limit = R).
// The searched point's "own id" is its index in the parent point set (start + index), the
// builder-side quirk of the reference; visited comes back sorted by (dist, id).
void build_search(const BuildView &V, const HostGraph &G, int32_t index, Scratch &S) {
  const int64_t B = V.L;
  const int bits = std::max<int>(10, (int)std::ceil(std::log2((double)(B * B))) - 2);
  const uint64_t mask = (UINT64_C(1) << bits) - 1;
  S.table.assign((size_t)1 << bits, -1);
  S.front.clear();
  S.visited.clear();
  const float *q = V.vec(index);
  const int64_t qid = V.start + index;
  S.front.push_back(mkkey(host_distance(V.metric, V.vec(0), q, (int)V.d), 0));
  size_t p = 0;
  int64_t nvis = 0;
  while (p < S.front.size() && nvis < V.n) {
    const uint64_t cur = S.front[p];
    S.front[p] = cur | 1;
    S.visited.push_back(cur & ~UINT64_C(1));
    nvis++;
    const int32_t *row = G.row(key_id(cur));
    const int64_t deg = std::min<int64_t>(row[0], G.maxdeg);
    const float cutoff = ((int64_t)S.front.size() < B) ? (float)INT_MAX : key_dist(S.front.back());
    S.cand.clear();
    for (int64_t i = 0; i < deg; i++) {
      const int32_t a = row[1 + i];
      if ((int64_t)a == qid) continue;
      const uint64_t loc = mix64((uint64_t)(int64_t)a) & mask;
      if (S.table[loc] == a) continue;
      S.table[loc] = a;
      const float dd = host_distance(V.metric, V.vec(a), q, (int)V.d);
      if (dd >= cutoff) continue;
      S.cand.push_back(mkkey(dd, a));
    }
    size_t first_ins = S.front.size();
    if (!S.cand.empty()) {
      std::sort(S.cand.begin(), S.cand.end());
      // in-place union from the back, dropping candidates already present
      std::vector<uint64_t> &F = S.front;
      size_t keepc = 0;
      for (size_t ci = 0; ci < S.cand.size(); ci++) {  // set_union: max(copies in F, copies in cand)
        const uint64_t c = S.cand[ci];
        size_t j = 0;
        while (j < ci && S.cand[ci - 1 - j] == c) j++;
        auto it = std::lower_bound(F.begin(), F.end(), c, [](uint64_t a, uint64_t b) { return (a | 1) < (b | 1); });
        size_t bx = 0;
        while (it + bx != F.end() && ((*(it + bx) | 1) == (c | 1))) bx++;
        if (j < bx) continue;
        S.keep.push_back(c);
        keepc++;
      }
      S.cand.swap(S.keep);
      S.keep.clear();
      if (keepc) {
        size_t m = F.size(), total = m + keepc;
        F.resize(total);
        size_t i = m, j = keepc, o = total;
        while (j > 0) {
          if (i > 0 && (F[i - 1] | 1) > (S.cand[j - 1] | 1)) F[--o] = F[--i];
          else {
            F[--o] = S.cand[--j];
            first_ins = o;
          }
        }
        if ((int64_t)F.size() > B) F.resize((size_t)B);
      }
    }
    size_t sp = std::min(p, first_ins);
    p = S.front.size();
    for (size_t x = sp; x < S.front.size(); x++)
      if (!(S.front[x] & 1)) {
        p = x;
        break;
      }
  }
  std::sort(S.visited.begin(), S.visited.end());
}

// robustPrune (vamana/index.h:61-108); cand = (key(dist to p, id)); ties by id
void robust_prune(const BuildView &V, const HostGraph &G, int32_t p, std::vector<uint64_t> &cand, bool add,
                  std::vector<int32_t> &out) {
  if (add) {
    const int32_t *row = G.row(p);
    for (int32_t i = 0; i < row[0]; i++) cand.push_back(mkkey(V.dist(row[1 + i], p), row[1 + i]));
  }
  std::sort(cand.begin(), cand.end());
  out.clear();
  const size_t nc = cand.size();
  std::vector<char> dead(nc, 0);
  size_t idx = 0;
  while ((int64_t)out.size() < V.R && idx < nc) {
    const size_t at = idx++;
    if (dead[at]) continue;
    const int32_t ps = key_id(cand[at]);
    if (ps == p) continue;
    out.push_back(ps);
    const float *vs = V.vec(ps);
    for (size_t i = idx; i < nc; i++) {
      if (dead[i]) continue;
      const float d_sp = host_distance(V.metric, vs, V.vec(key_id(cand[i])), (int)V.d);
      if (V.alpha * (double)d_sp <= (double)key_dist(cand[i])) dead[i] = 1;
    }
  }
}

}  // namespace

void vamana_build(const float *pts, int64_t stride, int64_t d, int metric, int64_t start, int64_t n,
                  int64_t R, int64_t L, double alpha, HostGraph &G, int threads) {
  BuildView V{pts, stride, d, metric, start, n, R, L, alpha};
  G.n = n;
  G.maxdeg = (int32_t)R;
  G.rows.assign((size_t)n * (R + 1), 0);
  if (n == 0) return;
  std::vector<int32_t> order((size_t)n);
  for (int64_t i = 0; i < n; i++) order[i] = (int32_t)i;
  std::sort(order.begin(), order.end(), [](int32_t a, int32_t b) { return mix64((uint64_t)a) < mix64((uint64_t)b); });
  size_t cap = std::min<size_t>((size_t)(0.02 * (double)(float)n), 1000000ul);  // vamana/index.h:224-226
  if (cap == 0) cap = (size_t)n;
  const size_t m = (size_t)n;
  size_t count = 0, inc = 0;
  const int nthr = std::max(1, threads);
  static thread_local Scratch tls_scratch;  // one per pool thread, reused across builds
  while (count < m) {
    size_t lo, hi;
    if (std::pow(2.0, (double)inc) <= (double)cap) {
      lo = (size_t)std::pow(2.0, (double)inc) - 1;
      hi = std::min((size_t)std::pow(2.0, (double)(inc + 1)), m) - 1;
      count = hi;
    } else {
      lo = count;
      hi = std::min(count + cap, m);
      count += cap;
    }
    const size_t bs = hi - lo;
    std::vector<std::vector<int32_t>> fresh(bs);
    parallel_for((int64_t)bs, nthr, [&](int64_t bi) {
      Scratch &S = tls_scratch;
      const int32_t index = order[lo + bi];
      build_search(V, G, index, S);
      std::vector<uint64_t> cand(S.visited);
      robust_prune(V, G, index, cand, true, fresh[bi]);
    });
    for (size_t bi = 0; bi < bs; bi++) {
      int32_t *row = G.row(order[lo + bi]);
      row[0] = (int32_t)fresh[bi].size();
      std::copy(fresh[bi].begin(), fresh[bi].end(), row + 1);
    }
    // reverse edges grouped by target, sources in batch order (vamana/index.h:277-306)
    std::vector<std::pair<int32_t, int32_t>> rev;
    for (size_t bi = 0; bi < bs; bi++)
      for (int32_t t : fresh[bi]) rev.emplace_back(t, order[lo + bi]);
    std::stable_sort(rev.begin(), rev.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
    std::vector<size_t> cuts;
    for (size_t i = 0; i < rev.size(); i++)
      if (i == 0 || rev[i].first != rev[i - 1].first) cuts.push_back(i);
    cuts.push_back(rev.size());
    parallel_for((int64_t)cuts.size() - 1, nthr, [&](int64_t gi) {
      const size_t b = cuts[gi], e = cuts[gi + 1];
      const int32_t tgt = rev[b].first;
      int32_t *row = G.row(tgt);
      if ((int64_t)(e - b) + row[0] <= R) {
        for (size_t i = b; i < e; i++) row[1 + row[0]++] = rev[i].second;
      } else {
        std::vector<uint64_t> cand;
        cand.reserve(e - b + row[0]);
        for (size_t i = b; i < e; i++) cand.push_back(mkkey(V.dist(rev[i].second, tgt), rev[i].second));
        std::vector<int32_t> out;
        robust_prune(V, G, tgt, cand, true, out);
        row[0] = (int32_t)out.size();
        std::copy(out.begin(), out.end(), row + 1);
      }
    });
    inc++;
  }
  parallel_for(n, nthr, [&](int64_t i) {  // final neighbour sort (vamana/index.h:131-134)
    int32_t *row = G.row(i);
    std::vector<uint64_t> nb((size_t)row[0]);
    for (int32_t j = 0; j < row[0]; j++) nb[j] = mkkey(V.dist(i, row[1 + j]), row[1 + j]);
    std::sort(nb.begin(), nb.end());
    for (int32_t j = 0; j < row[0]; j++) row[1 + j] = key_id(nb[j]);
  });
}

// ------------------------------------------------------------------------------------------------
// index layout
// ------------------------------------------------------------------------------------------------
static void obtain_graph(HostIndex &H, HostPart &P, int threads, bool keep) {
  const BuildSpec &s = H.spec;
  P.lo = *std::min_element(H.labels.begin() + P.start, H.labels.begin() + P.start + P.n);
  P.hi = *std::max_element(H.labels.begin() + P.start, H.labels.begin() + P.start + P.n);
  std::string fn = s.cache.empty() ? std::string() : graph_file_name(s, P.lo, P.hi, P.n);
  if (!fn.empty() && exists(fn)) {
    if (!keep) return;
    if (!graph_file_load(fn, P.g)) throw std::runtime_error("cannot read graph cache file " + fn);
    if (P.g.n != P.n) throw std::runtime_error("graph cache file has the wrong size: " + fn);
    return;
  }
  vamana_build(H.pts.data(), s.stride, s.d, s.metric, P.start, P.n, s.R, s.L, s.alpha, P.g, threads);
  if (!fn.empty() && !graph_file_save(fn, P.g)) throw std::runtime_error("cannot write graph cache file " + fn);
  if (!keep) {
    P.g.rows.clear();
    P.g.rows.shrink_to_fit();
  }
}

void build_host_index(HostIndex &H, const float *points, const float *labels, int shard, int nshards) {
  BuildSpec &s = H.spec;
  if (s.n <= 0 || s.d <= 0) throw std::runtime_error("empty point set");
  if (s.threads <= 0) s.threads = default_threads();
  s.stride = ((s.d * 4 + 63) / 64) * 16;  // 64-byte rows (point_range.h:39-44), zero padded
  H.sorted = (s.kind == 2 || s.kind == 3 || s.kind == 4);
  H.vamana_leaves = (s.kind == 1 || s.kind == 3 || s.kind == 4);
  const int64_t n = s.n;
  std::vector<int64_t> order((size_t)n);
  for (int64_t i = 0; i < n; i++) order[i] = i;
  if (H.sorted)  // canonical form of the reference's unstable argsort: stable by (label, id)
    std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return labels[a] < labels[b]; });
  H.pts.assign((size_t)n * s.stride, 0.f);
  H.labels.resize((size_t)n);
  H.decoding.resize((size_t)n);
  parallel_for(n, s.threads, [&](int64_t r) {
    memcpy(H.pts.data() + r * s.stride, points + order[r] * s.d, (size_t)s.d * 4);
    H.labels[r] = labels[order[r]];
    H.decoding[r] = (uint32_t)order[r];
  });

  std::vector<std::pair<int, int64_t>> todo;  // (level, idx)
  auto add_level = [&](size_t nb) {
    H.levels.emplace_back(nb);
    for (size_t b = 0; b < nb; b++) todo.emplace_back((int)H.levels.size() - 1, (int64_t)b);
  };
  switch (s.kind) {
    case 0: {  // PrefilterIndex: label argsort only
      H.fi_sorted.resize((size_t)n);
      for (int64_t i = 0; i < n; i++) H.fi_sorted[i] = (int32_t)i;
      std::stable_sort(H.fi_sorted.begin(), H.fi_sorted.end(), [&](int32_t a, int32_t b) { return labels[a] < labels[b]; });
      H.fv_sorted.resize((size_t)n);
      for (int64_t i = 0; i < n; i++) H.fv_sorted[i] = labels[H.fi_sorted[i]];
      return;
    }
    case 1: {
      add_level(1);
      H.levels[0][0].start = 0;
      H.levels[0][0].n = n;
      break;
    }
    case 2:
    case 3: {  // B-ary window search tree
      const size_t B = (size_t)s.split_factor;
      if (B < 2) throw std::runtime_error("split_factor must be at least 2");
      H.offsets.push_back({0, n});
      add_level(1);
      H.levels[0][0].start = 0;
      H.levels[0][0].n = n;
      while ((size_t)H.offsets.back()[1] > (size_t)(int64_t)s.cutoff) {
        const std::vector<int64_t> &prev = H.offsets.back();
        const size_t pnb = prev.size() - 1;
        std::vector<int64_t> off(pnb * B + 1);
        off.back() = n;
        for (size_t b = 0; b < pnb; b++) {
          const size_t ls = (size_t)prev[b], lsz = (size_t)(prev[b + 1] - prev[b]);
          const size_t large = (lsz + B - 1) / B, small = large - 1, nl = lsz - small * B;
          for (size_t i = 0; i < B; i++)
            off[b * B + i] = (int64_t)(i < nl ? ls + i * large : ls + nl * large + (i - nl) * small);
        }
        H.offsets.push_back(off);
        add_level(pnb * B);
        for (size_t b = 0; b < pnb * B; b++) {
          H.levels.back()[b].start = off[b];
          H.levels.back()[b].n = off[b + 1] - off[b];
        }
      }
      break;
    }
    case 4: {  // super tree: sizes in float arithmetic exactly as the reference computes them
      const float fs = (float)s.split_factor, fsh = (float)s.shift_factor;
      if (fs <= 1) throw std::runtime_error("split_factor must be greater than 1");
      if (fsh >= 1 || fsh <= 0) throw std::runtime_error("shift_factor must be between 0 and 1");
      add_level(1);
      H.levels[0][0].start = 0;
      H.levels[0][0].n = n;
      H.sup_size.push_back(n);
      H.sup_shift.push_back(0);
      while ((size_t)H.sup_size.back() > (size_t)(int64_t)s.cutoff) {
        const size_t last = (size_t)H.sup_size.back();
        const size_t bsz = (size_t)(((float)last + fs - 1) / fs);
        const size_t shift = (size_t)std::ceil((float)bsz * fsh);
        H.sup_size.push_back((int64_t)bsz);
        H.sup_shift.push_back((int64_t)shift);
        const size_t nb = (((size_t)n - bsz) + shift - 1) / shift + 1;
        add_level(nb);
        for (size_t b = 0; b < nb; b++) {
          const size_t bs = b * shift, be = std::min(bs + bsz, (size_t)n);
          H.levels.back()[b].start = (int64_t)bs;
          H.levels.back()[b].n = (int64_t)(be - bs);
        }
      }
      break;
    }
    default:
      throw std::runtime_error("unknown index kind");
  }
  if (!H.vamana_leaves) return;
  if (s.R > 64) throw std::runtime_error("max_degree > 64 is not supported by the gfx950 search kernel");

  // graphs: big partitions one after another with a parallel build inside, small ones in parallel
  const bool sharded = nshards > 0;
  std::vector<size_t> big, small;
  for (size_t t = 0; t < todo.size(); t++) {
    if (sharded && (int)(t % (size_t)nshards) != shard) continue;
    const HostPart &P = H.levels[todo[t].first][todo[t].second];
    (P.n >= 16384 ? big : small).push_back(t);
  }
  for (size_t t : big) obtain_graph(H, H.levels[todo[t].first][todo[t].second], s.threads, !sharded);
  std::mutex emu;
  std::string err;
  parallel_for((int64_t)small.size(), s.threads, [&](int64_t i) {
    try {
      size_t t = small[i];
      obtain_graph(H, H.levels[todo[t].first][todo[t].second], 1, !sharded);
    } catch (std::exception &e) {
      std::lock_guard<std::mutex> lk(emu);
      err = e.what();
    }
  });
  if (!err.empty()) throw std::runtime_error(err);
}

}  // namespace wann
