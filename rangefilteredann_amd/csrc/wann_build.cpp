// wann_build.cpp -- host-side index construction (see wann_build.h for the reference citations).
//
// The Vamana builder follows the reference's algorithm (prefix-doubling batches over a fixed
// insertion order, beam search on the snapshot, robustPrune, reverse edges appended or re-pruned,
// final per-node neighbour sort) with this engine's own data structures: the search frontier is
// one sorted array of 64-bit keys (order-preserving distance bits | node id | visited bit) merged
// in place, which is also the representation the gfx950 search kernel uses.  Insertion order is
// the reference's (parlay::random_permutation, insertion_order()) and distance ties break by id, so
// builds are deterministic for any thread count and the graph files equal the reference builder's
// whenever no two candidates of a prune / neighbour sort are exactly equidistant (with such ties
// the reference's order is whatever libstdc++'s std::sort leaves; ours is by id).
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
#include <chrono>

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

// uint8 / int8 rows (d bytes behind the float pointer): exact int32 sums cast to float, like the reference's
// (euclidian_point.h:44-60, mips_point.h:44-58) and like the device's v_dot4 routines
template <typename T>
static float byte_distance(int metric, const T *p, const T *q, int d) {
  int32_t r = 0;
  if (metric == 1) {
    for (int i = 0; i < d; i++) r += (int32_t)p[i] * (int32_t)q[i];
    return -(float)r;
  }
  for (int i = 0; i < d; i++) {
    const int32_t t = (int32_t)p[i] - (int32_t)q[i];
    r += t * t;
  }
  return (float)r;
}

// metric: bit 0 = inner product; bits 4-5 = element type of the rows (0 float32, 1 uint8, 2 int8)
float host_distance(int metric, const float *p, const float *q, int d) {
  const int dtype = (metric >> 4) & 3;
  if (dtype == 1) return byte_distance<uint8_t>(metric & 1, (const uint8_t *)p, (const uint8_t *)q, d);
  if (dtype == 2) return byte_distance<int8_t>(metric & 1, (const int8_t *)p, (const int8_t *)q, d);
  return (metric & 1) ? mips_ref_order(p, q, d) : l2_ref_order(p, q, d);
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
    int32_t kept[128 + 1];  // (max_degree <= 128: wann.h WANN_MAX_DEGREE)
    int nk = 0;
    for (int64_t i = 0; i < deg; i++) {
      const int32_t a = row[1 + i];
      if ((int64_t)a == qid) continue;
      const uint64_t loc = mix64((uint64_t)(int64_t)a) & mask;
      if (S.table[loc] == a) continue;
      S.table[loc] = a;
      kept[nk++] = a;
      const char *pv = (const char *)V.vec(a);  // the search is latency bound: get the misses in flight
      for (int64_t off = 0; off < V.stride * 4; off += 64) __builtin_prefetch(pv + off);
    }
    for (int i = 0; i < nk; i++) {
      const int32_t a = kept[i];
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
        const char *pr = (const char *)G.row(key_id(S.front[x]));
        for (int off = 0; off < (G.maxdeg + 1) * 4; off += 64) __builtin_prefetch(pr + off);
        break;
      }
  }
  std::sort(S.visited.begin(), S.visited.end());
}

// robustPrune (vamana/index.h:61-108); cand = (key(dist to p, id)); ties by id
// Order of exactly equidistant candidates.  The reference sorts by distance ONLY with std::sort (vamana/index.h:77-78,
// graph.h:106): equal keys end up wherever libstdc++'s introsort leaves them, a deterministic function of the input
// sequence.  The builders reproduce that -- the same std::sort on the same sequence (visited list in (dist, id) order,
// then the current out-neighbours; sources in batch order) with the same comparator; the GPU builder runs a restatement
// of the algorithm, wann_stdsort.h -- so that the graph files equal the reference's byte for byte also for
// integer-valued vectors (SIFT).  WANN_REF_TIES=0 switches to "ties break by id" (a test hook).
static bool ref_ties() {  // default on; WANN_REF_TIES=0: ties by id
  const char *e = getenv("WANN_REF_TIES");
  return !(e && *e == '0');
}
static inline bool dist_only_less(uint64_t a, uint64_t b) { return (a >> 32) < (b >> 32); }

// cand: the visited list of the build search (any order) when from_search, else the incoming sources in batch order
void robust_prune(const BuildView &V, const HostGraph &G, int32_t p, std::vector<uint64_t> &cand, bool add,
                  std::vector<int32_t> &out, bool from_search = false) {
  const bool ref = ref_ties();
  if (ref && from_search) std::sort(cand.begin(), cand.end());  // the reference keeps its visited list sorted by (dist, id)
  if (add) {
    const int32_t *row = G.row(p);
    for (int32_t i = 0; i < row[0]; i++) cand.push_back(mkkey(V.dist(row[1 + i], p), row[1 + i]));
  }
  if (ref) std::sort(cand.begin(), cand.end(), dist_only_less);
  else std::sort(cand.begin(), cand.end());
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

// Lock-step build of many partitions: round r inserts batch r of EVERY unfinished partition, so one
// parallel region covers ~2 % of all points of all partitions (hundreds of thousands of searches
// at SIFT-1M scale) instead of one partition's batch.  Each partition sees exactly the batch
// schedule of a stand-alone build, so the graphs do not depend on how partitions are grouped or
// on the thread count.
namespace {
struct Job {
  BuildView V;
  HostGraph *G;
  std::vector<int32_t> order;
  size_t cap = 0, count = 0, inc = 0, lo = 0, hi = 0;
  bool active = true;
  std::vector<std::vector<int32_t>> fresh;
  std::vector<std::pair<int32_t, int32_t>> rev;
  std::vector<size_t> cuts;
};
}  // namespace

// parlay::hash64 (parlay/utilities.h:131-141), the generator behind parlay::random
static inline uint64_t parlay_hash64(uint64_t u) {
  uint64_t v = u * UINT64_C(3935559000370003845) + UINT64_C(2691343689449507681);
  v ^= v >> 21;
  v ^= v << 37;
  v ^= v >> 4;
  v *= UINT64_C(4768777513237032717);
  v ^= v << 20;
  v ^= v >> 41;
  v ^= v << 5;
  return v;
}

// The order in which the reference inserts the n points of a graph: parlay::random_permutation<int>(n)
// with the default generator (vamana/index.h:233; parlay/random.h:79-159).  n < 8192: Fisher-Yates from
// the back with draws hash64(i); otherwise a stable bucketing of 0..n-1 by the low bits of hash64(i)
// followed by a Fisher-Yates pass per bucket with draws hash64(i + hash64(hash64(bucket))).  A function
// of n only, so builds are reproducible for any thread count and match the reference's graphs.
std::vector<int32_t> insertion_order(int64_t n) {
  std::vector<int32_t> ord((size_t)std::max<int64_t>(n, 0));
  auto shuffle = [](int32_t *a, size_t len, uint64_t salt) {
    for (size_t i = len; i-- > 1;) std::swap(a[i], a[parlay_hash64(i + salt) % (i + 1)]);
  };
  if (n < 8192) {
    for (int64_t i = 0; i < n; i++) ord[(size_t)i] = (int32_t)i;
    shuffle(ord.data(), ord.size(), 0);
    return ord;
  }
  int lg = 0;
  while (((uint64_t)1 << lg) < (uint64_t)n) lg++;  // ceil(log2 n)
  const int bits = ((uint64_t)n < ((uint64_t)1 << 27)) ? (lg - 7) / 2 : lg - 17;
  const uint64_t nb = (uint64_t)1 << bits;
  std::vector<size_t> first(nb + 1, 0);
  std::vector<uint32_t> bucket((size_t)n);
  for (int64_t i = 0; i < n; i++) first[(bucket[(size_t)i] = (uint32_t)(parlay_hash64((uint64_t)i) & (nb - 1))) + 1]++;
  for (uint64_t b = 0; b < nb; b++) first[b + 1] += first[b];
  {
    std::vector<size_t> fill(first.begin(), first.end() - 1);
    for (int64_t i = 0; i < n; i++) ord[fill[bucket[(size_t)i]]++] = (int32_t)i;
  }
  for (uint64_t b = 0; b < nb; b++) shuffle(ord.data() + first[b], first[b + 1] - first[b], parlay_hash64(parlay_hash64(b)));
  return ord;
}

static void build_many(std::vector<Job> &jobs, int threads) {
  const int nthr = std::max(1, threads);
  parallel_for((int64_t)jobs.size(), nthr, [&](int64_t j) {
    Job &J = jobs[j];
    const int64_t n = J.V.n;
    J.G->n = n;
    J.G->maxdeg = (int32_t)J.V.R;
    J.G->rows.assign((size_t)n * (J.V.R + 1), 0);
    J.order = insertion_order(n);
    J.cap = std::min<size_t>((size_t)(0.02 * (double)(float)n), 1000000ul);  // vamana/index.h:224-226
    if (J.cap == 0) J.cap = (size_t)n;
    J.active = n > 0;
  });
  static thread_local Scratch tls_scratch;  // one per pool thread, reused across builds
  std::vector<std::pair<int32_t, int32_t>> items;  // (job, index in batch) / (job, group)
  const bool verbose = getenv("WANN_VERBOSE") != nullptr;
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double tA = 0, tB = 0, tC = 0, t0 = now();
  size_t round_no = 0, total_items = 0;
  for (;;) {
    items.clear();
    for (size_t j = 0; j < jobs.size(); j++) {
      Job &J = jobs[j];
      if (!J.active) continue;
      const size_t m = (size_t)J.V.n;
      if (std::pow(2.0, (double)J.inc) <= (double)J.cap) {  // vamana/index.h:245-253
        J.lo = (size_t)std::pow(2.0, (double)J.inc) - 1;
        J.hi = std::min((size_t)std::pow(2.0, (double)(J.inc + 1)), m) - 1;
        J.count = J.hi;
      } else {
        J.lo = J.count;
        J.hi = std::min(J.count + J.cap, m);
        J.count += J.cap;
      }
      J.fresh.assign(J.hi - J.lo, {});
      for (size_t bi = 0; bi < J.hi - J.lo; bi++) items.emplace_back((int32_t)j, (int32_t)bi);
    }
    bool any_active = false;
    for (auto &J : jobs) any_active = any_active || J.active;
    if (!any_active) break;
    if (items.empty()) {  // every active partition has an empty batch this round (m = 1, or the step after 2^i - 1 == m - 1)
      for (auto &J : jobs) {
        if (!J.active) continue;
        J.inc++;
        if (J.count >= (size_t)J.V.n) J.active = false;
      }
      continue;
    }
    round_no++;
    total_items += items.size();
    double t1 = now();
    // phase A: search the snapshot + robustPrune, every insert of every partition's batch
    parallel_for((int64_t)items.size(), nthr, [&](int64_t it) {
      Job &J = jobs[items[it].first];
      const int32_t index = J.order[J.lo + items[it].second];
      Scratch &S = tls_scratch;
      build_search(J.V, *J.G, index, S);
      std::vector<uint64_t> cand(S.visited);
      robust_prune(J.V, *J.G, index, cand, true, J.fresh[items[it].second], true);
    });
    double t2 = now();
    tA += t2 - t1;
    // phase B: publish the batch's out-edges, group the reverse edges by target in batch order
    parallel_for((int64_t)jobs.size(), nthr, [&](int64_t j) {
      Job &J = jobs[j];
      if (!J.active) return;
      const size_t bs = J.hi - J.lo;
      for (size_t bi = 0; bi < bs; bi++) {
        int32_t *row = J.G->row(J.order[J.lo + bi]);
        row[0] = (int32_t)J.fresh[bi].size();
        std::copy(J.fresh[bi].begin(), J.fresh[bi].end(), row + 1);
      }
      J.rev.clear();
      for (size_t bi = 0; bi < bs; bi++)
        for (int32_t t : J.fresh[bi]) J.rev.emplace_back(t, J.order[J.lo + bi]);
      std::stable_sort(J.rev.begin(), J.rev.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
      J.cuts.clear();
      for (size_t i = 0; i < J.rev.size(); i++)
        if (i == 0 || J.rev[i].first != J.rev[i - 1].first) J.cuts.push_back(i);
      J.cuts.push_back(J.rev.size());
    });
    double t3 = now();
    tB += t3 - t2;
    // phase C: reverse edges -- append when the row has room, else robustPrune (vamana/index.h:297-306)
    items.clear();
    for (size_t j = 0; j < jobs.size(); j++)
      if (jobs[j].active)
        for (size_t g = 0; g + 1 < jobs[j].cuts.size(); g++) items.emplace_back((int32_t)j, (int32_t)g);
    parallel_for((int64_t)items.size(), nthr, [&](int64_t it) {
      Job &J = jobs[items[it].first];
      const size_t b = J.cuts[items[it].second], e = J.cuts[items[it].second + 1];
      const int32_t tgt = J.rev[b].first;
      int32_t *row = J.G->row(tgt);
      if ((int64_t)(e - b) + row[0] <= J.V.R) {
        for (size_t i = b; i < e; i++) row[1 + row[0]++] = J.rev[i].second;
      } else {
        std::vector<uint64_t> cand;
        cand.reserve(e - b + row[0]);
        for (size_t i = b; i < e; i++) cand.push_back(mkkey(J.V.dist(J.rev[i].second, tgt), J.rev[i].second));
        std::vector<int32_t> out;
        robust_prune(J.V, *J.G, tgt, cand, true, out);
        row[0] = (int32_t)out.size();
        std::copy(out.begin(), out.end(), row + 1);
      }
    });
    tC += now() - t3;
    if (verbose && (round_no % 10 == 0))
      fprintf(stderr, "[wann build] round %zu: %zu inserts so far, search+prune %.1fs, group %.1fs, reverse %.1fs\n", round_no,
              total_items, tA, tB, tC);
    for (auto &J : jobs) {
      if (!J.active) continue;
      J.inc++;
      if (J.count >= (size_t)J.V.n) {
        J.active = false;
        J.fresh.clear();
        J.fresh.shrink_to_fit();
        J.rev.clear();
        J.rev.shrink_to_fit();
      }
    }
  }
  if (verbose)
    fprintf(stderr, "[wann build] %zu partitions, %zu inserts, %zu rounds: search+prune %.1fs, group %.1fs, reverse %.1fs, total %.1fs (%d threads)\n",
            jobs.size(), total_items, round_no, tA, tB, tC, now() - t0, nthr);
  // final neighbour sort by distance to the node (vamana/index.h:131-134), all nodes of all partitions
  std::vector<int64_t> base(jobs.size() + 1, 0);
  for (size_t j = 0; j < jobs.size(); j++) base[j + 1] = base[j] + jobs[j].V.n;
  parallel_for(base.back(), nthr, [&](int64_t g) {
    size_t j = std::upper_bound(base.begin(), base.end(), g) - base.begin() - 1;
    Job &J = jobs[j];
    const int64_t i = g - base[j];
    int32_t *row = J.G->row(i);
    std::vector<uint64_t> nb((size_t)row[0]);
    for (int32_t x = 0; x < row[0]; x++) nb[x] = mkkey(J.V.dist(i, row[1 + x]), row[1 + x]);
    if (ref_ties()) std::sort(nb.begin(), nb.end(), dist_only_less);
    else std::sort(nb.begin(), nb.end());
    for (int32_t x = 0; x < row[0]; x++) row[1 + x] = key_id(nb[x]);
  });
}

void vamana_build(const float *pts, int64_t stride, int64_t d, int metric, int64_t start, int64_t n,
                  int64_t R, int64_t L, double alpha, HostGraph &G, int threads) {
  std::vector<Job> jobs(1);
  jobs[0].V = BuildView{pts, stride, d, metric, start, n, R, L, alpha};
  jobs[0].G = &G;
  build_many(jobs, threads);
}

// ------------------------------------------------------------------------------------------------
// index layout
// ------------------------------------------------------------------------------------------------
void build_host_index(HostIndex &H, const void *points, const float *labels, int shard, int nshards,
                      std::vector<HostPart *> *pending) {
  BuildSpec &s = H.spec;
  if (s.n <= 0 || s.d <= 0) throw std::runtime_error("empty point set");
  if (s.threads <= 0) s.threads = default_threads();
  const int64_t esz = s.dtype == 0 ? 4 : 1;     // bytes per element of the caller's rows and of the stored rows
  s.stride = ((s.d * esz + 63) / 64) * 16;      // 64-byte rows (point_range.h:39-44), zero padded; in 32-bit words
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
    memcpy(H.pts.data() + r * s.stride, (const char *)points + order[r] * s.d * esz, (size_t)(s.d * esz));
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
  if (s.R > 128) throw std::runtime_error("max_degree > 128 is not supported by the gfx950 search kernels (WANN_MAX_DEGREE)");

  // graphs: load what the cache holds, build the rest in lock-step, publish to the cache
  const bool sharded = nshards > 0;
  const bool keep = !sharded;
  std::vector<HostPart *> want;
  for (size_t t = 0; t < todo.size(); t++) {
    if (sharded && (int)(t % (size_t)nshards) != shard) continue;
    want.push_back(&H.levels[todo[t].first][todo[t].second]);
  }
  std::mutex emu;
  std::string err;
  std::vector<char> need(want.size(), 0);
  parallel_for((int64_t)want.size(), s.threads, [&](int64_t i) {
    try {
      HostPart &P = *want[i];
      P.lo = *std::min_element(H.labels.begin() + P.start, H.labels.begin() + P.start + P.n);
      P.hi = *std::max_element(H.labels.begin() + P.start, H.labels.begin() + P.start + P.n);
      const std::string fn = s.cache.empty() ? std::string() : graph_file_name(s, P.lo, P.hi, P.n);
      if (!fn.empty() && exists(fn)) {
        if (!keep) return;
        if (!graph_file_load(fn, P.g)) throw std::runtime_error("cannot read graph cache file " + fn);
        if (P.g.n != P.n) throw std::runtime_error("graph cache file has the wrong size: " + fn);
        return;
      }
      need[i] = 1;
    } catch (std::exception &e) {
      std::lock_guard<std::mutex> lk(emu);
      err = e.what();
    }
  });
  if (!err.empty()) throw std::runtime_error(err);
  std::vector<HostPart *> todo_build;
  for (size_t i = 0; i < want.size(); i++)
    if (need[i]) todo_build.push_back(want[i]);
  if (pending) {
    *pending = todo_build;
    return;
  }
  build_pending_on_host(H, todo_build);
  save_built_graphs(H, todo_build, keep);
}

void build_pending_on_host(HostIndex &H, std::vector<HostPart *> &pending) {
  const BuildSpec &s = H.spec;
  std::vector<Job> jobs;
  for (HostPart *P : pending) {
    Job J;
    J.V = BuildView{H.pts.data(), s.stride, s.d, s.metric | (s.dtype << 4), P->start, P->n, s.R, s.L, s.alpha};
    J.G = &P->g;
    jobs.push_back(std::move(J));
  }
  if (!jobs.empty()) build_many(jobs, s.threads);
}

void save_built_graphs(HostIndex &H, std::vector<HostPart *> &built, bool keep) {
  const BuildSpec &s = H.spec;
  if (s.cache.empty()) return;
  std::mutex emu;
  std::string err;
  parallel_for((int64_t)built.size(), s.threads, [&](int64_t i) {
    HostPart &P = *built[i];
    const std::string fn = graph_file_name(s, P.lo, P.hi, P.n);
    if (!graph_file_save(fn, P.g)) {
      std::lock_guard<std::mutex> lk(emu);
      err = "cannot write graph cache file " + fn;
    }
    if (!keep) {
      P.g.rows.clear();
      P.g.rows.shrink_to_fit();
    }
  });
  if (!err.empty()) throw std::runtime_error(err);
}

}  // namespace wann
