// wann_gpu_build.cpp -- lock-step GPU build of all missing partition graphs of an index
// (reference algorithm: ParlayANN/algorithms/vamana/index.h:211-313; the per-round structure is
// the one of build_many() in wann_build.cpp, with phases A-C running as gfx950 kernels).
#include "wann_gpu_build.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "wann_build_device.h"
#include "wann_hip_util.h"

namespace wann {

namespace {
struct JobState {
  std::vector<int32_t> order;
  size_t n = 0, cap = 0, count = 0, inc = 0, lo = 0, hi = 0;
  bool active = true;
};
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}  // namespace

void gpu_build_graphs(const IndexView &view, int32_t *d_graph, const std::vector<PartDesc> &parts,
                      std::vector<GpuBuildTarget> &targets, int64_t R, int64_t L, double alpha, int num_cus,
                      int threads, void *stream, int vis_scale) {
  if (targets.empty()) return;
  hipStream_t st = (hipStream_t)stream;
  // (re)start from empty rows: a retry after a visited-list overflow must not see the previous attempt
  for (auto &t : targets) {
    const PartDesc &pd = parts[t.part_index];
    HIP_CHECK(hipMemsetAsync(d_graph + pd.row_base * view.rs, 0xFF, (size_t)pd.n * view.rs * sizeof(int32_t), st));
  }
  const bool verbose = getenv("WANN_VERBOSE") != nullptr;
  const double t_begin = now_s();
  const int rs = view.rs;
  const size_t nj = targets.size();
  std::vector<JobState> jobs(nj);
  parallel_for((int64_t)nj, threads, [&](int64_t j) {
    JobState &J = jobs[j];
    const int64_t n = parts[targets[j].part_index].n;
    J.n = (size_t)n;
    J.order = insertion_order(n);
    J.cap = std::min<size_t>((size_t)(0.02 * (double)(float)n), 1000000ul);  // vamana/index.h:224-226
    if (J.cap == 0) J.cap = (size_t)n;
    J.active = n > 0;
  });
  size_t max_items = 0, max_group = 0;
  for (auto &J : jobs) {
    size_t b = std::max<size_t>(J.cap, 1);
    size_t p2 = 1;
    while (p2 * 2 <= std::max<size_t>(J.cap, 1)) p2 *= 2;  // the doubling batches never exceed cap either
    max_items += std::min(J.n, std::max(b, p2));
    max_group = std::max(max_group, std::min(J.n, std::max(b, p2)));
  }

  const int bits = std::max<int>(10, (int)std::ceil(std::log2((double)(L * L))) - 2);
  const int table_lds = bits <= kMaxLdsBits ? 1 : 0;
  int vis_cap = (int)(((2 * L * std::max(vis_scale, 1) + 64 + 63) / 64) * 64) + 64;
  if (const char *e = getenv("WANN_BUILD_VIS_CAP")) vis_cap = std::max(64, atoi(e)) * std::max(vis_scale, 1);  // tests: force the restart path
  const int wpb = build_waves_per_block();
  const int lds_insert = build_lds_bytes_per_wave(view.stride, (int)L, bits, table_lds, vis_cap, (int)R) * wpb;
  if (lds_insert > 160 * 1024) throw std::runtime_error("gpu build: build beam L too large for the LDS budget");
  int insert_blocks_per_cu = std::max(1, std::min(16 / wpb, (160 * 1024) / lds_insert));
  const int lds_rev = build_lds_bytes_per_wave(view.stride, 0, 0, 0, vis_cap, (int)R) * wpb;
  int rev_blocks_per_cu = std::max(1, std::min(16 / wpb, (160 * 1024) / lds_rev));
  const int max_insert_blocks = num_cus * insert_blocks_per_cu;
  const int max_rev_blocks = num_cus * rev_blocks_per_cu;
  const int big_blocks = std::max(1, num_cus / 2);
  const int64_t big_cap = (int64_t)max_group + R + 64;

  DevBuf<BuildItem> d_items;
  DevBuf<int32_t> d_fresh, d_fresh_cnt, d_ints, d_pair_val, d_sorted_val, d_gstart, d_fallback, d_gtable;
  DevBuf<unsigned long long> d_pair_key, d_sorted_key, d_big;
  DevBuf<unsigned char> d_temp;
  d_items.ensure(std::max<size_t>(max_items, 1));
  d_fresh.ensure(max_items * R);
  d_fresh_cnt.ensure(max_items);
  d_ints.ensure(16);
  d_pair_key.ensure(max_items * R);
  d_sorted_key.ensure(max_items * R);
  d_pair_val.ensure(max_items * R);
  d_sorted_val.ensure(max_items * R);
  d_gstart.ensure(max_items * R);
  d_fallback.ensure(max_items * R);
  if (!table_lds) d_gtable.ensure(((size_t)max_insert_blocks * wpb) << bits);
  d_big.ensure((size_t)big_blocks * wpb * big_cap);
  const size_t temp_bytes = build_sort_temp_bytes((int64_t)(max_items * R));
  d_temp.ensure(temp_bytes);
  int32_t *h_ints = nullptr;
  HIP_CHECK(hipHostMalloc((void **)&h_ints, 16 * sizeof(int32_t)));

  BuildArgs A{};
  {  // equidistant candidates in the order the reference's std::sort leaves them (wann_stdsort.h); WANN_REF_TIES=0: by id
    const char *e = getenv("WANN_REF_TIES");
    A.ref_ties = (e && *e == '0') ? 0 : 1;
  }
  A.ix = view;
  A.ix.graph = d_graph;
  A.graph_rw = d_graph;
  A.items = d_items.p;
  A.cursor = d_ints.p + 0;
  A.cursor2 = d_ints.p + 1;
  A.cursor3 = d_ints.p + 2;
  A.ngroups = d_ints.p + 3;
  A.nfallback = d_ints.p + 4;
  A.err = d_ints.p + 5;
  A.L = (int32_t)L;
  A.bits = bits;
  A.R = (int32_t)R;
  A.alpha = alpha;
  A.vis_cap = vis_cap;
  A.fresh = d_fresh.p;
  A.fresh_cnt = d_fresh_cnt.p;
  A.g_table = d_gtable.p;
  A.pair_key = d_pair_key.p;
  A.sorted_key = d_sorted_key.p;
  A.pair_val = d_pair_val.p;
  A.sorted_val = d_sorted_val.p;
  A.gstart = d_gstart.p;
  A.fallback = d_fallback.p;
  A.big_sb = d_big.p;
  A.big_cap = big_cap;

  std::vector<BuildItem> items;
  size_t rounds = 0, total_items = 0;
  try {
    for (;;) {
      items.clear();
      for (size_t j = 0; j < nj; j++) {
        JobState &J = jobs[j];
        if (!J.active) continue;
        const size_t m = J.n;
        if (std::pow(2.0, (double)J.inc) <= (double)J.cap) {  // vamana/index.h:245-253
          J.lo = (size_t)std::pow(2.0, (double)J.inc) - 1;
          J.hi = std::min((size_t)std::pow(2.0, (double)(J.inc + 1)), m) - 1;
          J.count = J.hi;
        } else {
          J.lo = J.count;
          J.hi = std::min(J.count + J.cap, m);
          J.count += J.cap;
        }
        for (size_t bi = J.lo; bi < J.hi; bi++) items.push_back(BuildItem{targets[j].part_index, J.order[bi]});
      }
      bool any_active = false;
      for (auto &J : jobs) any_active = any_active || J.active;
      if (!any_active) break;
      if (items.empty()) {  // every active partition has an empty batch this round
        for (auto &J : jobs) {
          if (!J.active) continue;
          J.inc++;
          if (J.count >= J.n) J.active = false;
        }
        continue;
      }
      if (items.size() > max_items) throw std::runtime_error("gpu build: internal batch bound exceeded");
      rounds++;
      total_items += items.size();
      HIP_CHECK(hipMemcpyAsync(d_items.p, items.data(), items.size() * sizeof(BuildItem), hipMemcpyHostToDevice, st));
      HIP_CHECK(hipMemsetAsync(d_ints.p, 0, 16 * sizeof(int32_t), st));
      A.nitems = (int32_t)items.size();
      A.npairs = (int64_t)items.size() * R;
      A.big = 0;
      int blocks = (int)std::min<int64_t>(max_insert_blocks, ((int64_t)items.size() + wpb - 1) / wpb);
      if (launch_build_insert(A, blocks, table_lds, st)) throw HipError(std::string("k_build_insert: ") + build_launch_last_error());
      if (launch_build_publish(A, st)) throw HipError(std::string("k_build_publish: ") + build_launch_last_error());
      if (launch_build_sort_groups(A, d_temp.p, temp_bytes, st)) throw HipError(std::string("build sort: ") + build_launch_last_error());
      blocks = (int)std::min<int64_t>(max_rev_blocks, (A.npairs + wpb - 1) / wpb);
      if (launch_build_reverse(A, blocks, st)) throw HipError(std::string("k_build_reverse: ") + build_launch_last_error());
      HIP_CHECK(hipMemcpyAsync(h_ints, d_ints.p, 16 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
      HIP_CHECK(hipStreamSynchronize(st));
      if (h_ints[5] & 1) throw std::runtime_error("gpu build overflow: a visited list exceeded the LDS candidate buffer");
      if (h_ints[4] > 0) {  // hub nodes whose reverse-edge group did not fit the LDS buffer
        A.big = 1;
        if (launch_build_reverse(A, std::min(big_blocks, (h_ints[4] + wpb - 1) / wpb), st))
          throw HipError(std::string("k_build_reverse(big): ") + build_launch_last_error());
        HIP_CHECK(hipMemcpyAsync(h_ints, d_ints.p, 16 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        if (h_ints[5] & 2) throw std::runtime_error("gpu build overflow: a reverse-edge group exceeded the global candidate buffer");
      }
      if (verbose && rounds % 10 == 0)
        fprintf(stderr, "[wann gpu build] round %zu: %zu inserts so far, %.1fs\n", rounds, total_items, now_s() - t_begin);
      for (auto &J : jobs) {
        if (!J.active) continue;
        J.inc++;
        if (J.count >= J.n) J.active = false;
      }
    }
    // final per-node neighbour sort: tiles of 64 rows
    items.clear();
    for (size_t j = 0; j < nj; j++)
      for (size_t r0 = 0; r0 < jobs[j].n; r0 += 64) items.push_back(BuildItem{targets[j].part_index, (int32_t)r0});
    DevBuf<BuildItem> d_tiles;
    d_tiles.upload(items);
    HIP_CHECK(hipMemsetAsync(d_ints.p, 0, 16 * sizeof(int32_t), st));
    A.items = d_tiles.p;
    A.nitems = (int32_t)items.size();
    int blocks = (int)std::min<int64_t>((int64_t)num_cus * (16 / wpb), ((int64_t)items.size() + wpb - 1) / wpb);
    if (launch_build_final(A, blocks, st)) throw HipError(std::string("k_build_final: ") + build_launch_last_error());
    HIP_CHECK(hipStreamSynchronize(st));
    const double t_dev = now_s();
    // copy the finished rows back in the reference's in-memory layout
    size_t max_rows = 0;
    for (auto &J : jobs) max_rows = std::max(max_rows, J.n);
    std::vector<int32_t> stage(max_rows * rs);
    for (size_t j = 0; j < nj; j++) {
      const PartDesc &pd = parts[targets[j].part_index];
      HostGraph &g = targets[j].part->g;
      g.n = pd.n;
      g.maxdeg = (int32_t)R;
      g.rows.assign((size_t)pd.n * (R + 1), 0);
      HIP_CHECK(hipMemcpy(stage.data(), d_graph + pd.row_base * rs, (size_t)pd.n * rs * 4, hipMemcpyDeviceToHost));
      parallel_for(pd.n, pd.n >= 65536 ? threads : 1, [&](int64_t i) {
        const int32_t *r = stage.data() + i * rs;
        int32_t *o = g.row(i);
        int deg = 0;
        while (deg < R && r[deg] >= 0) {
          o[1 + deg] = r[deg];
          deg++;
        }
        o[0] = deg;
      });
    }
    if (verbose)
      fprintf(stderr, "[wann gpu build] %zu partitions, %zu inserts, %zu rounds: device %.1fs, copy-back %.1fs\n", nj,
              total_items, rounds, t_dev - t_begin, now_s() - t_dev);
  } catch (...) {
    (void)hipHostFree(h_ints);
    throw;
  }
  (void)hipHostFree(h_ints);
}

}  // namespace wann
