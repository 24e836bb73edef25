// abi_sanitize_test.cpp -- second driver of `make sanitize`: the host side of the C ABI (wann_host.cpp, built with
// g++ -fsanitize=address,undefined) on a machine WITHOUT a GPU: argument validation, the loud no-device failures,
// error strings, and the host-only cache-shard build.  (With a GPU present only the validation paths run.)
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/wann.h"

static int fails = 0;
#define CHECK(c)                                                                                           \
  do {                                                                                                     \
    if (!(c)) {                                                                                            \
      fprintf(stderr, "CHECK failed at line %d: %s (last error: %s)\n", __LINE__, #c, wann_last_error()); \
      fails++;                                                                                             \
    }                                                                                                      \
  } while (0)

int main(int argc, char **argv) {
  const std::string tmp = argc > 1 ? argv[1] : "/tmp/wann_sanitize";
  CHECK(wann_abi_version() == WANN_ABI_VERSION);
  const int64_t n = 600, d = 10;
  std::vector<float> pts((size_t)n * d), labels((size_t)n);
  for (size_t i = 0; i < pts.size(); i++) pts[i] = (float)((i * 2654435761u) % 97);
  for (int64_t i = 0; i < n; i++) labels[(size_t)i] = (float)((i * 131) % n);
  const std::string cache = tmp + "/abi_";
  wann_build_params bp{8, 16, 1.0, cache.c_str()};
  // invalid arguments never reach the device
  CHECK(wann_index_create(9, 0, WANN_DTYPE_F32, pts.data(), n, d, labels.data(), 100, 2, 0.5, &bp, 0, 2) == nullptr);
  CHECK(strstr(wann_last_error(), "invalid argument") != nullptr);
  CHECK(wann_index_create(WANN_KIND_TREE_VAMANA, 0, 7, pts.data(), n, d, labels.data(), 100, 2, 0.5, &bp, 0, 2) == nullptr);
  CHECK(wann_index_create(WANN_KIND_TREE_VAMANA, 0, WANN_DTYPE_F32, nullptr, n, d, labels.data(), 100, 2, 0.5, &bp, 0, 2) == nullptr);
  wann_query_params qp{10, 20, 1.35, 1000000, 10000, 2, 10000, 0, 0.f, 0};
  uint32_t ids[10];
  float dists[10];
  CHECK(wann_batch_search(nullptr, pts.data(), labels.data(), 1, "fenwick", &qp, ids, dists) == WANN_ERR_INVALID);
  CHECK(wann_batch_search_device(nullptr, nullptr, nullptr, 0, 0, "fenwick", &qp, nullptr, nullptr, nullptr) == WANN_ERR_INVALID);
  wann_counters c;
  CHECK(wann_get_counters(nullptr, &c) == WANN_ERR_INVALID);
  CHECK(wann_num_points(nullptr) == -1 && wann_dim(nullptr) == -1 && wann_num_levels(nullptr) == -1 && wann_max_degree(nullptr) == -1);
  int64_t s, e;
  CHECK(wann_partition_range(nullptr, 0, 0, &s, &e) == WANN_ERR_INVALID);
  if (wann_device_count() == 0) {
    CHECK(wann_index_create(WANN_KIND_TREE_VAMANA, 0, WANN_DTYPE_F32, pts.data(), n, d, labels.data(), 100, 2, 0.5, &bp, 0, 2) == nullptr);
    CHECK(strstr(wann_last_error(), "no usable gfx950 device") != nullptr);
    std::vector<int32_t> rows((size_t)n * 9, 0), oi(16);
    std::vector<float> od(16);
    std::vector<int64_t> hops(2), cmps(2);
    CHECK(wann_raw_beam_search(0, pts.data(), n, d, rows.data(), 8, 0, n, pts.data(), 2, nullptr, 8, 1000, 64, oi.data(), od.data(), oi.data(),
                               hops.data(), cmps.data(), 0) == WANN_ERR_NO_DEVICE);
  }
  // host-only graph cache build, both shards, the graph kinds and both metrics; uint8 rows take the byte conversion path
  for (int kind : {WANN_KIND_POSTFILTER, WANN_KIND_TREE_VAMANA, WANN_KIND_SUPER})
    for (int shard = 0; shard < 2; shard++)
      CHECK(wann_build_cache_shard(kind, shard, WANN_DTYPE_F32, pts.data(), n, d, labels.data(), 100, 2, 0.5, &bp, shard, 2, 2) == WANN_OK);
  std::vector<uint8_t> bytes((size_t)n * d);
  for (size_t i = 0; i < bytes.size(); i++) bytes[i] = (uint8_t)(i * 37);
  CHECK(wann_build_cache_shard(WANN_KIND_TREE_VAMANA, 0, WANN_DTYPE_U8, bytes.data(), n, d, labels.data(), 100, 2, 0.5, &bp, 0, 1, 2) == WANN_OK);
  CHECK(wann_build_cache_shard(WANN_KIND_TREE_VAMANA, 0, WANN_DTYPE_F32, pts.data(), n, d, labels.data(), 100, 2, 0.5, &bp, 3, 2, 2) == WANN_ERR_INVALID);
  wann_build_params nocache{8, 16, 1.0, ""};
  CHECK(wann_build_cache_shard(WANN_KIND_TREE_VAMANA, 0, WANN_DTYPE_F32, pts.data(), n, d, labels.data(), 100, 2, 0.5, &nocache, 0, 1, 2) == WANN_ERR_INVALID);
  if (fails) {
    fprintf(stderr, "abi sanitize test: %d check(s) failed\n", fails);
    return 1;
  }
  printf("ABI_SANITIZE_OK\n");
  return 0;
}
