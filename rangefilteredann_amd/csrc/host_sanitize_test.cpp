// host_sanitize_test.cpp -- driver of the `make sanitize` target: the CPU-side C++ of the engine (index layout,
// label sort, window-search-tree / super-tree shapes, the host Vamana builder, graph cache I/O, the
// insertion permutation) under AddressSanitizer + UndefinedBehaviorSanitizer.  CPU build only: GPU
// sanitizers are not available on the pool, and nothing here touches HIP.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <unistd.h>
#include <vector>

#include "../../include/wann.h"
#include "wann_build.h"
#include "wann_stdsort.h"
#include <algorithm>

using namespace wann;

static int fails = 0;
#define CHECK(c)                                                       \
  do {                                                                 \
    if (!(c)) {                                                        \
      fprintf(stderr, "CHECK failed at line %d: %s\n", __LINE__, #c);  \
      fails++;                                                         \
    }                                                                  \
  } while (0)

static void check_graph(const HostGraph &g, int64_t n, int64_t R) {
  CHECK(g.n == n && g.maxdeg == R);
  for (int64_t i = 0; i < g.n; i++) {
    const int32_t *r = g.row(i);
    CHECK(r[0] >= 0 && r[0] <= g.maxdeg);
    for (int j = 0; j < r[0]; j++) CHECK(r[1 + j] >= 0 && r[1 + j] < n);
  }
}

int main(int argc, char **argv) {
  const std::string tmp = argc > 1 ? argv[1] : "/tmp/wann_sanitize";
  (void)system(("mkdir -p " + tmp).c_str());
  std::mt19937 rng(7);
  std::normal_distribution<float> nd;
  for (int metric = 0; metric < 2; metric++)
    for (int kind : {WANN_KIND_PREFILTER, WANN_KIND_POSTFILTER, WANN_KIND_TREE_PREFILTER, WANN_KIND_TREE_VAMANA, WANN_KIND_SUPER}) {
      const int64_t n = kind == WANN_KIND_SUPER ? 1500 : 2300, d = metric ? 20 : 13;  // d not a multiple of 4 / 8 / 16
      std::vector<float> pts((size_t)n * d), labels((size_t)n);
      for (auto &x : pts) x = metric ? nd(rng) : std::rint(nd(rng) * 20.f);  // integer-valued rows: distance ties
      for (int64_t i = 0; i < n; i++) labels[(size_t)i] = (float)((i * 7919) % n) + 0.5f;
      for (int pass = 0; pass < 2; pass++) {  // pass 0 builds and saves the cache, pass 1 loads it
        HostIndex H;
        H.spec.kind = kind;
        H.spec.metric = metric;
        H.spec.n = n;
        H.spec.d = d;
        H.spec.cutoff = 300;
        H.spec.split_factor = kind == WANN_KIND_SUPER ? 2.5 : 3;
        H.spec.shift_factor = 0.3;
        H.spec.R = 12;
        H.spec.L = 24;
        H.spec.alpha = 1.1;
        H.spec.threads = 3;
        H.spec.cache = tmp + "/k" + std::to_string(kind) + "m" + std::to_string(metric) + "_";
        std::vector<HostPart *> pending;
        build_host_index(H, pts.data(), labels.data(), -1, 0, &pending);
        CHECK(pass == 0 || pending.empty());
        if (!pending.empty()) {
          build_pending_on_host(H, pending);
          save_built_graphs(H, pending, true);
        }
        CHECK((int64_t)H.labels.size() == n && (int64_t)H.pts.size() == n * H.spec.stride);
        for (auto &lv : H.levels)
          for (auto &P : lv) {
            CHECK(P.start >= 0 && P.start + P.n <= n);
            if (H.vamana_leaves) check_graph(P.g, P.n, H.spec.R);
          }
      }
      // the sharded cache build (wann_build_cache_shard's body) writes the same files
      for (int shard = 0; shard < 2; shard++) {
        HostIndex H;
        H.spec.kind = kind;
        H.spec.metric = metric;
        H.spec.n = n;
        H.spec.d = d;
        H.spec.cutoff = 300;
        H.spec.split_factor = kind == WANN_KIND_SUPER ? 2.5 : 3;
        H.spec.shift_factor = 0.3;
        H.spec.R = 12;
        H.spec.L = 24;
        H.spec.alpha = 1.1;
        H.spec.threads = 2;
        H.spec.cache = tmp + "/shard_k" + std::to_string(kind) + "m" + std::to_string(metric) + "_";
        build_host_index(H, pts.data(), labels.data(), shard, 2);
      }
    }
  // graph file round trip, truncated / corrupt files must be rejected without reading out of bounds
  {
    HostGraph g;
    std::vector<float> pts(400 * 16);
    for (auto &x : pts) x = nd(rng);
    vamana_build(pts.data(), 16, 9, 0, 0, 400, 8, 16, 1.2, g, 2);
    check_graph(g, 400, 8);
    const std::string path = tmp + "/roundtrip.bin";
    CHECK(graph_file_save(path, g));
    HostGraph h;
    CHECK(graph_file_load(path, h) && h.n == g.n && h.maxdeg == g.maxdeg);
    for (int64_t i = 0; i < g.n && h.n == g.n; i++) {  // (slots past the degree are unspecified in a built row)
      CHECK(h.row(i)[0] == g.row(i)[0]);
      for (int j = 1; j <= g.row(i)[0]; j++) CHECK(h.row(i)[j] == g.row(i)[j]);
    }
    CHECK(truncate(path.c_str(), 100) == 0);
    HostGraph t;
    CHECK(!graph_file_load(path, t));
    CHECK(!graph_file_load(tmp + "/does_not_exist.bin", t));
  }
  for (int64_t n : {1, 2, 3, 100, 8191, 8192, 8193, 20000}) {
    std::vector<int32_t> p = insertion_order(n);
    std::vector<char> seen((size_t)n, 0);
    CHECK((int64_t)p.size() == n);
    for (int32_t x : p) {
      CHECK(x >= 0 && x < n && !seen[(size_t)x]);
      if (x >= 0 && x < n) seen[(size_t)x] = 1;
    }
  }
  // the restated std::sort (wann_stdsort.h) against libstdc++'s own, distance-only comparator, sequences full of ties
  {
    auto check = [&](std::vector<uint64_t> v) {
      std::vector<uint64_t> want = v, got = v;
      std::sort(want.begin(), want.end(), DistOnlyLess());
      int32_t stack[128];
      std_sort_emulated(got.data(), (int)got.size(), stack, DistOnlyLess());
      CHECK(got == want);
      want = v;
      got = v;
      std::sort(want.begin(), want.end(), FullKeyLess());
      std_sort_emulated(got.data(), (int)got.size(), stack, FullKeyLess());
      CHECK(got == want);
    };
    for (int n = 0; n <= 2600; n += (n < 70 ? 1 : 37)) {
      for (int levels : {1, 2, 5, 40, 1 << 20}) {  // number of distinct distances
        std::vector<uint64_t> v((size_t)n);
        for (int i = 0; i < n; i++) v[(size_t)i] = ((uint64_t)(rng() % (uint32_t)levels) << 32) | ((uint64_t)(uint32_t)i << 1);
        check(v);
        std::sort(v.begin(), v.end());
        check(v);  // already sorted
        std::reverse(v.begin(), v.end());
        check(v);
      }
    }
    // a median-of-three killer (exhausts the depth limit: the heapsort branch) and an organ pipe
    for (int n : {64, 512, 2048, 4096}) {
      std::vector<uint64_t> v((size_t)n);
      const int k = n / 2;
      for (int i = 1; i <= k; i++) {
        if (i & 1) {
          v[(size_t)i - 1] = (uint64_t)i << 32;
          v[(size_t)i] = (uint64_t)(k + i) << 32;
        }
        v[(size_t)(k + i - 1)] = (uint64_t)(2 * i) << 32;
      }
      for (int i = 0; i < n; i++) v[(size_t)i] |= (uint64_t)(uint32_t)i << 1;
      check(v);
      for (int i = 0; i < n; i++) v[(size_t)i] = ((uint64_t)(i < n / 2 ? i : n - i) << 32) | ((uint64_t)(uint32_t)i << 1);
      check(v);
    }
  }
  // the C ABI's argument validation (no device needed for these paths) lives in wann_host.cpp and is covered by tests/test_abi.py
  if (fails) {
    fprintf(stderr, "host sanitize test: %d check(s) failed\n", fails);
    return 1;
  }
  printf("HOST_SANITIZE_OK\n");
  return 0;
}
