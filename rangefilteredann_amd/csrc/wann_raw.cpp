// wann_raw.cpp -- one graph over one contiguous slice of a point set, resident on the device: wann_raw_beam_search (tests,
// micro-benchmarks: every search core through one entry point) and the unfiltered VamanaIndex API (wann_vamana_*).
#include "wann_host_internal.h"

extern "C" {

// One graph over one contiguous slice of a point set, resident on the device: the object behind wann_raw_beam_search
// (tests, micro-benchmarks) and behind the unfiltered VamanaIndex API.
struct RawGraph {
  wann_index I;  // scratch object: config_for / device properties
  DevBuf<float> d_pts;
  DevBuf<int32_t> d_rows;
  DevBuf<PartDesc> d_parts;
  int64_t n = 0, d = 0, subset_n = 0;
  int32_t maxdeg = 0;
  // per-call buffers, kept across calls (a VamanaIndex answers batch after batch)
  DevBuf<float> d_q, d_rd;
  DevBuf<int32_t> d_list, d_ints, d_rid, d_rsz, g_table, g_epoch;
  DevBuf<Task> d_tasks;
  DevBuf<long long> d_hops, d_cmps, d_qids;
  DevBuf<Counters> d_ctr;
  DevBuf<unsigned long long> g_beam, d_prof;
  DevBuf<uint32_t> g_seen;
  int64_t layout = -1;
  // points: (n, d) rows of `dtype` elements (float32, or uint8 / int8 bytes: stored as byte rows)
  void load(int device, int metric, const void *points, int64_t n_, int64_t d_, const int32_t *graph_rows, int64_t maxdeg_,
            int64_t subset_start, int64_t subset_n_, int dtype = WANN_DTYPE_F32) {
    HIP_CHECK(hipSetDevice(device));
    I.device = device;
    I.tune = Tuning::from_env();
    hipDeviceProp_t prop;
    HIP_CHECK(hipGetDeviceProperties(&prop, device));
    I.num_cus = prop.multiProcessorCount;
    n = n_;
    d = d_;
    subset_n = subset_n_;
    maxdeg = (int32_t)maxdeg_;
    const int64_t esz = dtype == WANN_DTYPE_F32 ? 4 : 1;
    const int64_t stride = ((d * esz + 63) / 64) * 16;  // 32-bit words per row
    std::vector<float> pts((size_t)n * stride, 0.f);
    for (int64_t i = 0; i < n; i++) memcpy(pts.data() + i * stride, (const char *)points + i * d * esz, (size_t)(d * esz));
    const int rs = (int)(((maxdeg_ + 15) / 16) * 16);
    HostGraph g;
    g.n = subset_n;
    g.maxdeg = (int32_t)maxdeg_;
    g.rows.assign(graph_rows, graph_rows + (size_t)subset_n * (maxdeg_ + 1));
    std::vector<int32_t> rows((size_t)subset_n * rs);
    convert_rows(g, rs, rows.data());
    d_pts.upload(pts);
    d_rows.upload(rows);
    std::vector<PartDesc> parts{{0, (int32_t)subset_start, (int32_t)subset_n}};
    d_parts.upload(parts);
    I.view.points = d_pts.p;
    I.view.graph = d_rows.p;
    I.view.parts = d_parts.p;
    I.view.labels = d_pts.p;  // unused in raw mode
    I.view.n = n;
    I.view.d = (int32_t)d;
    I.view.stride = (int32_t)stride;
    I.view.rs = rs;
    I.view.maxdeg = (int32_t)maxdeg_;
    I.view.metric = metric;
    I.view.dtype = dtype;
  }
  // one beam search per query (host buffers); cut_k > 0: the k / cut step of beamSearch.h:159-167 (first-generation core)
  void search(const float *queries, int64_t nq, const int64_t *query_ids, int64_t beam, int64_t limit, int64_t degree_limit,
              int64_t cut_k, double cut, int32_t *out_ids, float *out_dists, int32_t *out_sizes, int64_t *out_hops, int64_t *out_dist_cmps) {
    HIP_CHECK(hipSetDevice(I.device));
    std::vector<float> qv(queries, queries + (size_t)nq * d);
    d_q.upload(qv);
    std::vector<Task> tasks((size_t)nq);
    std::vector<int32_t> list((size_t)nq);
    std::vector<long long> qids((size_t)nq);
    for (int64_t i = 0; i < nq; i++) {
      tasks[i] = Task{(int32_t)i, T_GRAPH, 0, 0, 0, 0, 0.f, 0.f};
      list[i] = (int32_t)i;
      qids[i] = query_ids ? query_ids[i] : i;
    }
    d_tasks.upload(tasks);
    d_list.upload(list);
    d_qids.upload(qids);
    std::vector<int32_t> ints{(int32_t)nq, 0, 0, 0};
    d_ints.upload(ints);
    d_rid.ensure((size_t)nq * beam);
    d_rd.ensure((size_t)nq * beam);
    d_rsz.ensure(nq);
    d_hops.ensure(nq);
    d_cmps.ensure(nq);
    d_ctr.ensure(1);
    HIP_CHECK(hipMemset(d_ctr.p, 0, sizeof(Counters)));
    const bool with_cut = cut_k > 0;
    if (I.tune.hooks_live) I.tune = Tuning::from_env();  // (tests flip the core switches between calls on one VamanaIndex)
    const Tuning &T = I.tune;
    const bool wide = I.view.rs > 64;
    const bool old_general = T.old_general || with_cut || wide, force_general = T.force_general || with_cut || wide;
    // (dev / test switches: the large-LDS one-wave configuration; the first-generation cores live in that kernel only)
    const int raw_pool = kSearchPoolBytes;
    RoundCfg rc = config_for(I, T, beam, beam, nq, T.raw_big_lds || old_general, force_general, old_general, raw_pool);
    SearchArgs sa{};
    sa.ix = I.view;
    sa.queries = d_q.p;
    sa.tasks = d_tasks.p;
    sa.list = d_list.p;
    sa.list_count = d_ints.p;
    sa.cursor = d_ints.p + 1;
    sa.B = (int32_t)beam;
    sa.cap_inkernel = (int32_t)beam;
    sa.max_beam = INT32_MAX;
    sa.mult = 1;
    sa.pool_bytes = rc.pool_bytes;
    sa.force_general = force_general ? 1 : 0;
    sa.k = 1;
    sa.limit = limit;
    sa.degree_limit = (int32_t)std::min<int64_t>(degree_limit, INT32_MAX);
    sa.ctr = d_ctr.p;
    sa.raw = 1;
    sa.raw_ids = d_rid.p;
    sa.raw_dists = d_rd.p;
    sa.raw_sizes = d_rsz.p;
    sa.raw_hops = d_hops.p;
    sa.raw_cmps = d_cmps.p;
    sa.raw_qids = d_qids.p;
    sa.cut_k = (int32_t)cut_k;
    sa.cut = cut;
    sa.old_general = old_general ? 1 : 0;
    sa.helper = (rc.lc.big == 1 && T.helper) ? kHelpers : 0;
    if (rc.table_bits) {
      const int64_t seen_words = ((subset_n + 127) / 128) * 4;
      ensure_filter_scratch(g_table, g_epoch, g_seen, layout, rc.slots, rc.table_bits, seen_words, nullptr);
      sa.g_table = g_table.p;
      sa.g_table_bits = rc.table_bits;
      sa.g_epoch = g_epoch.p;
      sa.g_seen = g_seen.p;
      sa.g_seen_words = seen_words;
    }
    if (rc.beam_cap) {
      sa.g_beam_cap = rc.beam_cap;
      g_beam.ensure((size_t)rc.slots * sa.g_beam_cap);
      sa.g_beam = g_beam.p;
    }
    const bool prof = T.profile_phases;
    if (prof) {
      d_prof.ensure(16);
      HIP_CHECK(hipMemset(d_prof.p, 0, 16 * sizeof(unsigned long long)));
      sa.prof = d_prof.p;
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const bool verbose = T.verbose;
    if (verbose) {
      HIP_CHECK(hipEventCreate(&e0));
      HIP_CHECK(hipEventCreate(&e1));
      HIP_CHECK(hipEventRecord(e0, nullptr));
    }
    if (launch_search(sa, rc.lc, nullptr)) throw HipError(std::string("k_search: ") + launch_last_error());
    if (verbose) HIP_CHECK(hipEventRecord(e1, nullptr));
    HIP_CHECK(hipDeviceSynchronize());
    if (verbose) {
      float ms = 0.f;
      HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
      fprintf(stderr, "[wann raw] beam %ld nq %ld kernel kind %d blocks %d (%d per CU by the runtime's occupancy): %.3f ms\n", (long)beam, (long)nq, rc.lc.big, rc.lc.blocks,
              search_occupancy(sa, rc.lc), ms);
      (void)hipEventDestroy(e0);
      (void)hipEventDestroy(e1);
    }
    if (prof) {
      unsigned long long h[16];
      HIP_CHECK(hipMemcpy(h, d_prof.p, sizeof h, hipMemcpyDeviceToHost));
      fprintf(stderr, "[wann phases] beam=%ld nq=%ld cycles: row %llu filter %llu dist %llu merge %llu next %llu (built with make PROFILE=1?)\n", (long)beam,
              (long)nq, h[0], h[1], h[2], h[3], h[4]);
      fprintf(stderr, "[wann phases 5..8] %llu %llu %llu %llu (second-generation core: select / row+probes / next+requests / slot test / filter / "
                      "next packet / distances / delta insert / truncation); probe wait %llu, flush %llu\n", h[5], h[6], h[7], h[8], h[9], h[10]);
    }
    HIP_CHECK(hipMemcpy(out_ids, d_rid.p, (size_t)nq * beam * 4, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(out_dists, d_rd.p, (size_t)nq * beam * 4, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(out_sizes, d_rsz.p, (size_t)nq * 4, hipMemcpyDeviceToHost));
    std::vector<long long> hh((size_t)nq), cc((size_t)nq);
    HIP_CHECK(hipMemcpy(hh.data(), d_hops.p, (size_t)nq * 8, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(cc.data(), d_cmps.p, (size_t)nq * 8, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < nq; i++) {
      if (out_hops) out_hops[i] = hh[i];
      if (out_dist_cmps) out_dist_cmps[i] = cc[i];
    }
  }
};

int wann_raw_beam_search(int metric, const float *points, int64_t n, int64_t d, const int32_t *graph_rows,
                         int64_t maxdeg, int64_t subset_start, int64_t subset_n, const float *queries, int64_t nq,
                         const int64_t *query_ids, int64_t beam, int64_t limit, int64_t degree_limit,
                         int32_t *out_ids, float *out_dists, int32_t *out_sizes, int64_t *out_hops,
                         int64_t *out_dist_cmps, int device) {
  if (usable_devices() <= device || device < 0)
    return fail(WANN_ERR_NO_DEVICE, "no usable gfx950 device (this library has no CPU search path)");
  if (maxdeg > WANN_MAX_DEGREE) return fail(WANN_ERR_UNSUPPORTED, "max_degree > 128 is not supported");
  try {
    RawGraph G;
    G.load(device, metric, points, n, d, graph_rows, maxdeg, subset_start, subset_n);
    G.search(queries, nq, query_ids, beam, limit, degree_limit, 0, 0.0, out_ids, out_dists, out_sizes, out_hops, out_dist_cmps);
  } catch (HipError &e) {
    return fail(WANN_ERR_HIP, e.what());
  } catch (std::exception &e) {
    return fail(WANN_ERR_INVALID, e.what());
  }
  return WANN_OK;
}

// ---- unfiltered VamanaIndex (ParlayANN/python/vamana_index.cpp:42-76, ParlayANN/python/builder.cpp) ----------------
namespace {
// point file: uint32 n, uint32 d, then n * d elements (point_range.h:63-93)
void read_point_file(const char *path, int dtype, std::vector<float> &out, int64_t &n, int64_t &d, std::vector<unsigned char> *raw_out = nullptr) {
  FILE *f = fopen(path, "rb");
  if (!f) throw std::runtime_error(std::string("cannot open point file ") + path);
  uint32_t head[2];
  if (fread(head, 4, 2, f) != 2) {
    fclose(f);
    throw std::runtime_error(std::string("point file too short: ") + path);
  }
  n = head[0];
  d = head[1];
  const size_t cnt = (size_t)n * d, esz = dtype == WANN_DTYPE_F32 ? 4 : 1;
  std::vector<unsigned char> raw(cnt * esz);
  const size_t got = cnt ? fread(raw.data(), esz, cnt, f) : 0;
  fclose(f);
  if (got != cnt) throw std::runtime_error(std::string("point file truncated: ") + path);
  if (dtype == WANN_DTYPE_F32) {
    out.resize(cnt);
    memcpy(out.data(), raw.data(), cnt * 4);
  } else
    out = bytes_to_float(dtype, raw.data(), (int64_t)cnt);
  if (raw_out) raw_out->swap(raw);
}
}  // namespace

struct wann_vamana {
  RawGraph G;
  int dtype = WANN_DTYPE_F32;
  std::mutex mu;
};

wann_vamana *wann_vamana_open(int metric, int dtype, const char *data_path, const char *graph_path, int device) {
  if ((metric != 0 && metric != 1) || dtype < 0 || dtype > 2 || !data_path || !graph_path) {
    fail(WANN_ERR_INVALID, "invalid argument to wann_vamana_open");
    return nullptr;
  }
  if (usable_devices() <= device || device < 0) {
    fail(WANN_ERR_NO_DEVICE, "no usable gfx950 device (this library has no CPU search path)");
    return nullptr;
  }
  try {
    std::unique_ptr<wann_vamana> V(new wann_vamana);
    V->dtype = dtype;
    std::vector<float> pts;
    std::vector<unsigned char> raw;
    int64_t n = 0, d = 0;
    read_point_file(data_path, dtype, pts, n, d, &raw);
    HostGraph g;
    if (!graph_file_load(graph_path, g)) throw std::runtime_error(std::string("cannot read graph file ") + graph_path);
    if (g.n != n) throw std::runtime_error("graph file and point file disagree on the number of points");
    if (g.maxdeg > WANN_MAX_DEGREE) throw std::runtime_error("max_degree > 128 is not supported");
    V->G.load(device, metric, raw.data(), n, d, g.rows.data(), g.maxdeg, 0, n, dtype);
    return V.release();
  } catch (HipError &e) {
    fail(WANN_ERR_HIP, e.what());
  } catch (std::exception &e) {
    fail(WANN_ERR_IO, e.what());
  }
  return nullptr;
}

void wann_vamana_close(wann_vamana *v) { delete v; }
int64_t wann_vamana_num_points(const wann_vamana *v) { return v ? v->G.n : -1; }
int64_t wann_vamana_dim(const wann_vamana *v) { return v ? v->G.d : -1; }

int wann_vamana_batch_search(wann_vamana *V, const void *queries, int64_t nq, int64_t knn, int64_t beam, uint32_t *ids, float *dists) {
  if (!V || nq < 0 || knn <= 0 || beam <= 0 || (nq > 0 && (!queries || !ids || !dists)))
    return fail(WANN_ERR_INVALID, "invalid argument to wann_vamana_batch_search");
  if (beam < knn) return fail(WANN_ERR_INVALID, "beam_width must be at least knn (the reference reads past its beam otherwise)");
  std::lock_guard<std::mutex> lk(V->mu);
  try {
    if (nq == 0) return WANN_OK;
    std::vector<float> qf;
    if (V->dtype != WANN_DTYPE_F32) {
      qf = bytes_to_float(V->dtype, queries, nq * V->G.d);
      queries = qf.data();
    }
    std::vector<int32_t> bid((size_t)nq * beam), bsz((size_t)nq);
    std::vector<float> bd((size_t)nq * beam);
    // QueryParams(knn, beam_width, 1.35, G.size(), G.max_degree()) (vamana_index.cpp:56); query i carries id i (:66)
    V->G.search((const float *)queries, nq, nullptr, beam, V->G.n, V->G.maxdeg, knn, 1.35, bid.data(), bd.data(), bsz.data(), nullptr, nullptr);
    for (int64_t i = 0; i < nq; i++)
      for (int64_t j = 0; j < knn; j++) {
        const bool have = j < bsz[(size_t)i];  // (the reference reads past a shorter beam: defined here as id 2^32-1, FLT_MAX)
        ids[i * knn + j] = have ? (uint32_t)bid[(size_t)(i * beam + j)] : 0xFFFFFFFFu;
        dists[i * knn + j] = have ? bd[(size_t)(i * beam + j)] : 3.402823466e+38f;
      }
  } catch (HipError &e) {
    return fail(WANN_ERR_HIP, e.what());
  } catch (std::exception &e) {
    return fail(WANN_ERR_INVALID, e.what());
  }
  return WANN_OK;
}

int wann_vamana_build_file(int metric, int dtype, const char *data_path, const char *graph_out_path, int64_t max_degree, int64_t limit,
                           double alpha, int device) {
  if ((metric != 0 && metric != 1) || dtype < 0 || dtype > 2 || !data_path || !graph_out_path)
    return fail(WANN_ERR_INVALID, "invalid argument to wann_vamana_build_file");
  try {
    std::vector<float> pts;
    int64_t n = 0, d = 0;
    std::vector<unsigned char> raw;  // (byte point sets are built from their bytes: exact integer distances)
    read_point_file(data_path, dtype, pts, n, d, &raw);
    if (n <= 0 || d <= 0) return fail(WANN_ERR_INVALID, "empty point file");

    // one Vamana graph over the points in file order = the stand-alone post-filter index's graph
    // (knn_index::build_index, vamana/index.h:123-313, BuildParams(R, L, alpha) types.h:94).  The labels only have to be
    // distinct and increasing for the builder to keep file order: float(i) is that below 2^24 points
    if (n > ((int64_t)1 << 24)) return fail(WANN_ERR_UNSUPPORTED, "wann_vamana_build_file: more than 2^24 points are not supported");
    std::vector<float> labels((size_t)n);
    for (int64_t i = 0; i < n; i++) labels[(size_t)i] = (float)i;
    wann_build_params bp{max_degree, limit, alpha, ""};
    wann_index *I = wann_index_create(WANN_KIND_POSTFILTER, metric, dtype, dtype == WANN_DTYPE_F32 ? (const void *)pts.data() : (const void *)raw.data(), n, d,
                                      labels.data(), 1000, 2, 0.5, &bp, device, 0);
    if (!I) return WANN_ERR_HIP;  // (message already set)
    const HostGraph &g = I->H.levels[0][0].g;
    const bool ok = g.n == n && graph_file_save(graph_out_path, g);
    wann_index_destroy(I);
    if (!ok) return fail(WANN_ERR_IO, std::string("cannot write graph file ") + graph_out_path);
  } catch (HipError &e) {
    return fail(WANN_ERR_HIP, e.what());
  } catch (std::exception &e) {
    return fail(WANN_ERR_IO, e.what());
  }
  return WANN_OK;
}

}  // extern "C"
