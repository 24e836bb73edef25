// wann_abi.cpp -- the C ABI of include/wann.h: index life cycle, the blocking / asynchronous / in-process multi-device search
// calls, the RCCL all-gather of per-shard rows, cost prediction, introspection.  Every compute entry point needs a gfx950 device
// and fails loudly otherwise: there is no CPU search path in this library.
#include "wann_host_internal.h"

// One batch in flight beside the others: its own workspace, streams and worker thread.  The worker runs the same run_batch
// the blocking call runs (host-side waits included) -- on ITS stream, so the kernels of two consecutive batches share the GPU:
// while batch i's last searches finish, batch i+1 is routed and its first workgroups take the compute units that fall free.
struct wann_index::AsyncLane {
  Workspace ws;
  hipStream_t stream = nullptr, side = nullptr;
  hipEvent_t ready = nullptr;  // the caller's inputs (recorded on the caller's stream at submission)
  wann_counters last{};
  struct Job {
    const float *q, *r;
    int64_t nq, base;
    std::string method;
    wann_query_params qp;
    uint32_t *ids;
    float *dists;
    int64_t ticket;
    Tuning tune;  // the index's switches as they were at submission (the worker never reads the index's record)
  };
  std::thread th;
  std::mutex m;
  std::condition_variable cv;
  bool has_job = false, busy = false, stop = false;
  Job job{};
  int64_t finished = -1;  // ticket of the last finished job; its outcome:
  int rc = WANN_OK;
  std::string err;
  void loop(wann_index *I) {
    for (;;) {
      Job j;
      {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return has_job || stop; });
        if (stop) return;
        j = job;
        has_job = false;
      }
      int code = WANN_OK;
      std::string msg;
      try {
        HIP_CHECK(hipSetDevice(I->device));
        HIP_CHECK(hipStreamWaitEvent(stream, ready, 0));
        run_batch(*I, ws, side, last, j.q, j.r, j.nq, j.base, j.method.c_str(), j.qp, j.ids, j.dists, stream, j.tune);
      } catch (HipError &e) {
        code = WANN_ERR_HIP;
        msg = e.what();
      } catch (std::exception &e) {
        code = WANN_ERR_INVALID;
        msg = e.what();
      }
      {
        std::lock_guard<std::mutex> lk(m);
        rc = code;
        err = msg;
        finished = j.ticket;
        busy = false;
      }
      cv.notify_all();
    }
  }
  ~AsyncLane() {
    {
      std::lock_guard<std::mutex> lk(m);
      stop = true;
    }
    cv.notify_all();
    if (th.joinable()) th.join();
    if (stream) (void)hipStreamDestroy(stream);
    if (side) (void)hipStreamDestroy(side);
    if (ready) (void)hipEventDestroy(ready);
  }
};

// RCCL, opened with dlopen at first use: a host that never gathers on the device never loads it.
struct wann_index::Rccl {
  void *lib = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  std::vector<ncclComm_t> comms;
  void open(const std::vector<int> &devices) {
    lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!lib) throw std::runtime_error(std::string("cannot open librccl.so: ") + dlerror());
    auto sym = [&](const char *n) {
      void *p = dlsym(lib, n);
      if (!p) throw std::runtime_error(std::string("librccl.so lacks ") + n);
      return p;
    };
    CommInitAll = (decltype(CommInitAll))sym("ncclCommInitAll");
    CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
    AllGather = (decltype(AllGather))sym("ncclAllGather");
    GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
    GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
    GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
    comms.assign(devices.size(), nullptr);
    check(CommInitAll(comms.data(), (int)devices.size(), devices.data()), "ncclCommInitAll");
  }
  void check(ncclResult_t r, const char *what) const {
    if (r != ncclSuccess) throw HipError(std::string(what) + ": " + (GetErrorString ? GetErrorString(r) : "RCCL error"));
  }
  ~Rccl() {
    if (CommDestroy)
      for (ncclComm_t c : comms)
        if (c) (void)CommDestroy(c);
    // (the library stays loaded: its teardown at dlclose is not worth the risk at process exit)
  }
};

wann_index::wann_index() = default;

wann_index::~wann_index() {
  rccl.reset();
  lanes.clear();  // (joins the workers before the streams and buffers they use go away)
  if (own_stream) (void)hipStreamDestroy(own_stream);
  if (side_stream) (void)hipStreamDestroy(side_stream);
}

extern "C" {

int wann_abi_version(void) { return WANN_ABI_VERSION; }
const char *wann_last_error(void) { return g_err.c_str(); }
int wann_device_count(void) { return usable_devices(); }

wann_index *wann_index_create(int kind, int metric, int dtype, const void *points, int64_t n, int64_t d,
                              const float *labels, int32_t cutoff, double split_factor, double shift_factor,
                              const wann_build_params *bp, int device, int build_threads) {
  if (dtype != WANN_DTYPE_F32 && dtype != WANN_DTYPE_U8 && dtype != WANN_DTYPE_I8) {
    fail(WANN_ERR_INVALID, "unknown dtype");
    return nullptr;
  }
  if (kind < 0 || kind > 4 || (metric != 0 && metric != 1) || !points || !labels || n <= 0 || d <= 0) {
    fail(WANN_ERR_INVALID, "invalid argument to wann_index_create");
    return nullptr;
  }
  if (n >= (int64_t)1 << 31) {
    fail(WANN_ERR_UNSUPPORTED, "point sets of 2^31 or more rows are not supported");
    return nullptr;
  }
  // uint8 / int8 point sets (euclidian_point.h:44-60, mips_point.h:44-58: int32 accumulation, cast to float) are kept as
  // BYTE rows on the device and scored with v_dot4 into exact int32 sums: any dimension, a quarter of the vector traffic.
  if (usable_devices() <= device || device < 0) {
    fail(WANN_ERR_NO_DEVICE, "no usable gfx950 device (this library has no CPU search path)");
    return nullptr;
  }
  std::unique_ptr<wann_index> I(new wann_index);
  try {
    I->device = device;
    I->dtype = dtype;
    I->tune = Tuning::from_env();
    I->H.spec = make_spec(kind, metric, dtype, n, d, cutoff, split_factor, shift_factor, bp, build_threads);
    std::vector<HostPart *> pending;
    build_host_index(I->H, points, labels, -1, 0, &pending);
    // WANN_DEVICES=a,b,...: the index is replicated on every listed device and wann_batch_search (host buffers) cuts its batch
    // into contiguous shards, one per replica.  `device` is the primary if it is listed, else the first entry is.
    std::vector<int> extra;
    if (const char *dv = getenv("WANN_DEVICES")) {
      std::vector<int> list;
      for (const char *c = dv; *c;) {
        char *end = nullptr;
        const long v = strtol(c, &end, 10);
        if (end == c) break;
        list.push_back((int)v);
        c = (*end == ',') ? end + 1 : end;
      }
      for (int v : list)
        if (v < 0 || v >= usable_devices()) throw std::runtime_error("WANN_DEVICES names device " + std::to_string(v) + ", which does not exist");
      if (!list.empty()) {
        size_t prim = 0;
        for (size_t i = 0; i < list.size(); i++)
          if (list[i] == device) {
            prim = i;
            break;
          }
        I->device = list[prim];
        for (size_t i = 0; i < list.size(); i++)
          if (i != prim) extra.push_back(list[i]);
      }
    }
    upload_index(*I);
    if (!pending.empty()) build_pending(*I, pending);
    for (int dv : extra) {  // (after the build: the graphs are in the host index by now)
      std::unique_ptr<wann_index> R(new wann_index);
      R->Hp = &I->H;
      R->device = dv;
      R->dtype = dtype;
      R->tune = I->tune;
      upload_index(*R);
      I->replicas.push_back(std::move(R));
    }
    HIP_CHECK(hipSetDevice(I->device));
  } catch (HipError &e) {
    fail(WANN_ERR_HIP, e.what());
    return nullptr;
  } catch (std::exception &e) {
    fail(WANN_ERR_INVALID, e.what());
    return nullptr;
  }
  return I.release();
}

void wann_index_destroy(wann_index *index) { delete index; }

int wann_batch_search_device(wann_index *I, const void *d_queries, const float *d_ranges, int64_t nq,
                             int64_t query_id_base, const char *method, const wann_query_params *qp,
                             uint32_t *d_ids, float *d_dists, void *hip_stream) {
  if (!I || !qp || nq < 0) return fail(WANN_ERR_INVALID, "invalid argument to wann_batch_search_device");
  std::lock_guard<std::mutex> lk(I->mu);
  try {
    // NULL = the HIP default stream: ordered after everything the caller queued on its default stream
    // (torch's current stream unless changed), so freshly produced inputs / recycled output blocks are safe
    hipStream_t st = (hipStream_t)hip_stream;
    const Tuning T = snapshot_tuning(*I);  // (WANN_TEST_HOOKS=1 only: tests flip switches between batches)
    run_batch(*I, I->ws, I->side_stream, I->last, (const float *)d_queries, d_ranges, nq, query_id_base, method, *qp, d_ids, d_dists, st, T);
  } catch (HipError &e) {
    return fail(WANN_ERR_HIP, e.what());
  } catch (std::exception &e) {
    return fail(WANN_ERR_INVALID, e.what());
  }
  return WANN_OK;
}

int wann_batch_search_device_ids(wann_index *I, const void *d_queries, const float *d_ranges, int64_t nq, const int64_t *d_query_ids,
                                 const char *method, const wann_query_params *qp, uint32_t *d_ids, float *d_dists, void *hip_stream) {
  if (!I || !qp || nq < 0 || (nq > 0 && !d_query_ids)) return fail(WANN_ERR_INVALID, "invalid argument to wann_batch_search_device_ids");
  std::lock_guard<std::mutex> lk(I->mu);
  try {
    const Tuning T = snapshot_tuning(*I);
    run_batch(*I, I->ws, I->side_stream, I->last, (const float *)d_queries, d_ranges, nq, 0, method, *qp, d_ids, d_dists, (hipStream_t)hip_stream, T, d_query_ids);
  } catch (HipError &e) {
    return fail(WANN_ERR_HIP, e.what());
  } catch (std::exception &e) {
    return fail(WANN_ERR_INVALID, e.what());
  }
  return WANN_OK;
}

// Asynchronous form of the device-buffer call (wann.h): tickets are served by the index's lanes in turn -- two by default,
// WANN_ASYNC_LANES (1 .. 4, read when the first asynchronous call creates them) for deeper pipelines: a lane is a workspace, two
// streams and a worker thread, and the batches in flight share the GPU.
static int async_lane_count() {
  const char *v = getenv("WANN_ASYNC_LANES");
  const int n = v ? atoi(v) : 2;
  return n < 1 ? 1 : n > 4 ? 4 : n;
}

int wann_batch_search_device_async(wann_index *I, const void *d_queries, const float *d_ranges, int64_t nq, int64_t query_id_base,
                                   const char *method, const wann_query_params *qp, uint32_t *d_ids, float *d_dists, void *after_stream,
                                   int64_t *ticket) {
  if (!I || !qp || nq < 0 || !ticket) return fail(WANN_ERR_INVALID, "invalid argument to wann_batch_search_device_async");
  try {
    // One submission at a time; lanes_mu (which wann_wait takes too) is only held to look at / publish the lane table and the
    // ticket counter, never while this call waits for its lane to fall idle.
    std::lock_guard<std::mutex> sub(I->submit_mu);
    HIP_CHECK(hipSetDevice(I->device));
    const Tuning tune = snapshot_tuning(*I);
    int64_t t;
    wann_index::AsyncLane *Lp;
    {
      std::lock_guard<std::mutex> lk(I->lanes_mu);
      if (I->lanes.empty()) {
        // (all lanes or none: a lane table that a failed creation left half filled would be indexed out of bounds by odd tickets)
        std::vector<std::unique_ptr<wann_index::AsyncLane>> fresh;
        const int nl = async_lane_count();
        for (int l = 0; l < nl; l++) {
          std::unique_ptr<wann_index::AsyncLane> L(new wann_index::AsyncLane);
          int prio_low = 0, prio_high = 0;
          HIP_CHECK(hipDeviceGetStreamPriorityRange(&prio_low, &prio_high));
          HIP_CHECK(hipStreamCreateWithFlags(&L->stream, hipStreamNonBlocking));
          HIP_CHECK(hipStreamCreateWithPriority(&L->side, hipStreamNonBlocking, prio_high));
          HIP_CHECK(hipEventCreateWithFlags(&L->ready, hipEventDisableTiming));
          wann_index::AsyncLane *lp = L.get();
          L->th = std::thread([lp, I] { lp->loop(I); });
          fresh.push_back(std::move(L));
        }
        I->lanes.swap(fresh);
      }
      t = I->next_ticket;
      Lp = I->lanes[(size_t)(t % (int64_t)I->lanes.size())].get();
    }
    wann_index::AsyncLane &L = *Lp;
    {
      std::unique_lock<std::mutex> ll(L.m);
      L.cv.wait(ll, [&] { return !L.busy; });  // (ticket t - <number of lanes> has finished; wann_wait it BEFORE submitting this one to see its outcome)
      HIP_CHECK(hipEventRecord(L.ready, (hipStream_t)after_stream));
      L.job = wann_index::AsyncLane::Job{(const float *)d_queries, d_ranges, nq, query_id_base, method ? method : "", *qp, d_ids, d_dists, t, tune};
      L.has_job = true;
      L.busy = true;
    }
    {  // the ticket exists from here on (a submission that failed above took none)
      std::lock_guard<std::mutex> lk(I->lanes_mu);
      I->next_ticket = t + 1;
    }
    L.cv.notify_all();
    *ticket = t;
  } catch (HipError &e) {
    return fail(WANN_ERR_HIP, e.what());
  } catch (std::exception &e) {
    return fail(WANN_ERR_INVALID, e.what());
  }
  return WANN_OK;
}

int wann_wait(wann_index *I, int64_t ticket, wann_counters *out) {
  if (!I || ticket < 0) return fail(WANN_ERR_INVALID, "invalid argument to wann_wait");
  wann_index::AsyncLane *L = nullptr;
  {
    std::lock_guard<std::mutex> lk(I->lanes_mu);
    if (I->lanes.empty() || ticket >= I->next_ticket) return fail(WANN_ERR_INVALID, "wann_wait: no such ticket");
    L = I->lanes[(size_t)(ticket % (int64_t)I->lanes.size())].get();
  }
  std::unique_lock<std::mutex> ll(L->m);
  L->cv.wait(ll, [&] { return L->finished >= ticket; });
  if (L->finished != ticket) return fail(WANN_ERR_INVALID, "wann_wait: the ticket's lane has served a later ticket since (wait for ticket t before submitting t + 2)");
  if (out) *out = L->last;
  if (L->rc != WANN_OK) return fail(L->rc, L->err);
  return WANN_OK;
}

// Shard `shard` of `world` contiguous shards of an nq-query batch (the cut of wann_batch_search's multi-device mode and of
// wann_batch_search_allgather): first row, row count, and the common plane capacity.
int wann_gather_layout(int64_t nq, int world, int shard, int64_t *lo, int64_t *count, int64_t *cap) {
  if (nq < 0 || world <= 0 || shard < 0 || shard >= world) return fail(WANN_ERR_INVALID, "invalid argument to wann_gather_layout");
  const int64_t base = nq / world, rem = nq % world;
  if (lo) *lo = shard * base + std::min<int64_t>(shard, rem);
  if (count) *count = base + (shard < rem ? 1 : 0);
  if (cap) *cap = base + (rem ? 1 : 0);
  return WANN_OK;
}

// The in-process multi-device call with DEVICE-RESIDENT, gathered result rows: every replica searches its shard into its send
// planes and ONE ncclAllGather (RCCL over xGMI, a communicator per replica, one group call) leaves all shards' planes on every
// replica's device.
int wann_batch_search_allgather(wann_index *I, const void *queries, const float *ranges, int64_t nq, const char *method,
                                const wann_query_params *qp, int32_t **d_planes, int64_t *cap_out) {
  if (!I || !qp || nq < 0 || !d_planes || !cap_out || (nq > 0 && (!queries || !ranges)))
    return fail(WANN_ERR_INVALID, "invalid argument to wann_batch_search_allgather");
  if (qp->k <= 0 || qp->k > 1024) return fail(WANN_ERR_INVALID, "k must be in [1, 1024]");
  const int G = 1 + (int)I->replicas.size();
  std::vector<wann_index *> reps{I};
  for (auto &R : I->replicas) reps.push_back(R.get());
  std::vector<int> devs;
  for (wann_index *T : reps) devs.push_back(T->device);
  for (int a = 0; a < G; a++)
    for (int b = a + 1; b < G; b++)
      if (devs[a] == devs[b]) return fail(WANN_ERR_UNSUPPORTED, "wann_batch_search_allgather needs DISTINCT devices in WANN_DEVICES (one RCCL rank per device)");
  // (one call at a time: the communicators are created lazily, and the planes of a replica's workspace are read by the collective
  // after that replica's own mutex has been released)
  std::lock_guard<std::mutex> gather_lock(I->gather_mu);
  try {
    if (!I->rccl) {
      std::unique_ptr<wann_index::Rccl> r(new wann_index::Rccl);
      r->open(devs);
      I->rccl = std::move(r);
    }
    const int64_t k = qp->k, d = I->H.spec.d, esz = I->dtype == WANN_DTYPE_F32 ? 4 : 1;
    int64_t cap = 0;
    wann_gather_layout(nq, G, 0, nullptr, nullptr, &cap);
    if (cap == 0) cap = 1;
    std::vector<std::thread> threads;
    std::vector<int> codes((size_t)G, WANN_OK);
    std::vector<std::string> errs((size_t)G);
    for (int g = 0; g < G; g++) {
      wann_index *T = reps[(size_t)g];
      int64_t lo = 0, cnt = 0;
      wann_gather_layout(nq, G, g, &lo, &cnt, nullptr);
      threads.emplace_back([=, &codes, &errs] {
        try {
          std::lock_guard<std::mutex> lk(T->mu);
          HIP_CHECK(hipSetDevice(T->device));
          const Tuning tune = snapshot_tuning(*T);
          Workspace &W = T->ws;
          hipStream_t st = T->own_stream;
          W.gat_send.ensure((size_t)(2 * cap * k));
          W.gat_recv.ensure((size_t)((int64_t)G * 2 * cap * k));
          W.q_stage.ensure((size_t)std::max<int64_t>(cnt * d, 1));
          W.r_stage.ensure((size_t)std::max<int64_t>(cnt * 2, 1));
          std::vector<float> qf;
          const void *qsrc = (const char *)queries + lo * d * esz;
          if (cnt && T->dtype != WANN_DTYPE_F32) {
            qf = bytes_to_float(T->dtype, qsrc, cnt * d);
            qsrc = qf.data();
          }
          if (cnt) {
            HIP_CHECK(hipMemcpyAsync(W.q_stage.p, qsrc, (size_t)cnt * d * 4, hipMemcpyHostToDevice, st));
            HIP_CHECK(hipMemcpyAsync(W.r_stage.p, ranges + 2 * lo, (size_t)cnt * 8, hipMemcpyHostToDevice, st));
          }
          int32_t *ids_plane = W.gat_send.p, *dist_plane = W.gat_send.p + cap * k;
          // rows beyond this shard's count: the padding of the reference's result rows (id 0 / FLT_MAX)
          if (cnt < cap) {
            HIP_CHECK(hipMemsetAsync(ids_plane + cnt * k, 0, (size_t)((cap - cnt) * k) * 4, st));
            HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)(dist_plane + cnt * k), 0x7f7fffff, (size_t)((cap - cnt) * k), st));
          }
          run_batch(*T, W, T->side_stream, T->last, W.q_stage.p, W.r_stage.p, cnt, lo, method, *qp, (uint32_t *)ids_plane, (float *)dist_plane, st, tune);
        } catch (HipError &e) {
          codes[(size_t)g] = WANN_ERR_HIP;
          errs[(size_t)g] = e.what();
        } catch (std::exception &e) {
          codes[(size_t)g] = WANN_ERR_INVALID;
          errs[(size_t)g] = e.what();
        }
      });
    }
    for (auto &t : threads) t.join();
    for (int g = 0; g < G; g++)
      if (codes[(size_t)g] != WANN_OK) return fail(codes[(size_t)g], "replica " + std::to_string(g) + ": " + errs[(size_t)g]);
    // one all-gather of the [2][cap][k] planes, all replicas in one group call
    wann_index::Rccl &N = *I->rccl;
    N.check(N.GroupStart(), "ncclGroupStart");
    {
      // (an open group is always closed: a failing call between the two would leave the communicators unusable)
      struct GroupGuard {
        wann_index::Rccl &n;
        bool open = true;
        ~GroupGuard() {
          if (open) (void)n.GroupEnd();
        }
      } guard{N};
      for (int g = 0; g < G; g++) {
        wann_index *T = reps[(size_t)g];
        HIP_CHECK(hipSetDevice(T->device));
        N.check(N.AllGather(T->ws.gat_send.p, T->ws.gat_recv.p, (size_t)(2 * cap * k), ncclInt32, N.comms[(size_t)g], T->own_stream), "ncclAllGather");
      }
      guard.open = false;
      N.check(N.GroupEnd(), "ncclGroupEnd");
    }
    for (int g = 0; g < G; g++) {
      wann_index *T = reps[(size_t)g];
      HIP_CHECK(hipSetDevice(T->device));
      HIP_CHECK(hipStreamSynchronize(T->own_stream));
      d_planes[g] = T->ws.gat_recv.p;
    }
    HIP_CHECK(hipSetDevice(I->device));
    *cap_out = cap;
  } catch (HipError &e) {
    return fail(WANN_ERR_HIP, e.what());
  } catch (std::exception &e) {
    return fail(WANN_ERR_INVALID, e.what());
  }
  return WANN_OK;
}

// Predicted work of every query of a batch (wann.h): the batch is routed on the device without speculative levels and
// k_task_cost prices each query's tasks.  A planning call (one small launch pair and a copy), not part of the search.
int wann_predict_costs(wann_index *I, const float *ranges, int64_t nq, const char *method, const wann_query_params *qp, float *cost) {
  if (!I || !qp || nq < 0 || (nq > 0 && (!ranges || !cost))) return fail(WANN_ERR_INVALID, "invalid argument to wann_predict_costs");
  if (nq == 0) return WANN_OK;
  std::lock_guard<std::mutex> lk(I->mu);
  try {
    HIP_CHECK(hipSetDevice(I->device));
    Workspace &W = I->ws;
    const int mcode = method_code(method);
    const bool tree = I->host().spec.kind == WANN_KIND_TREE_PREFILTER || I->host().spec.kind == WANN_KIND_TREE_VAMANA;
    const bool single = !tree || (mcode == M_OPTIMIZED && !qp->has_min_query_to_bucket_ratio && I->host().spec.split_factor <= 4);
    const int maxt = single ? 1 : 96;
    const int k = (int)std::max<int64_t>(1, std::min<int64_t>(qp->k, 1024));
    W.ensure(nq, k, maxt, 0);
    W.r_stage.ensure((size_t)nq * 2);
    W.dist_stage.ensure((size_t)nq);
    const Tuning tune = snapshot_tuning(*I);
    hipStream_t st = I->own_stream;
    HIP_CHECK(hipMemcpyAsync(W.r_stage.p, ranges, (size_t)nq * 8, hipMemcpyHostToDevice, st));
    HIP_CHECK(hipMemsetAsync(W.ints.p, 0, kInts * sizeof(int32_t), st));
    HIP_CHECK(hipMemsetAsync(W.ctr.p, 0, sizeof(Counters), st));
    RouteArgs ra{};
    ra.ix = I->view;
    ra.ranges = W.r_stage.p;
    ra.nq = nq;
    ra.method = mcode;
    ra.maxt = maxt;
    ra.qtask_cnt = W.qtask_cnt.p;
    ra.k = k;
    ra.beam = (int32_t)std::min<int64_t>(std::max<int64_t>(qp->beam_width, 1), INT32_MAX);
    ra.max_beam = (int32_t)std::min<int64_t>(qp->postfiltering_max_beam, INT32_MAX);
    ra.has_ratio = qp->has_min_query_to_bucket_ratio;
    ra.ratio = qp->min_query_to_bucket_ratio;
    ra.tasks = W.tasks.p;
    ra.graph_list = W.list_a.p;
    ra.graph_count = W.ints.p + I_GRAPH_COUNT;
    ra.heavy_list = W.list_heavy.p;
    ra.heavy_count = W.ints.p + I_HEAVY_COUNT;
    ra.heavy_cap = W.big_stride;
    ra.mid_list = W.list_mid.p;
    ra.mid_count = W.ints.p + I_MID_COUNT;
    ra.heavy_ratio = kHeavyRatio;
    ra.risk_count = W.ints.p + I_RISK;
    ra.brute_list = W.list_brute.p;
    ra.brute_count = W.ints.p + I_BRUTE_COUNT;
    ra.spec = 0;  // (plain tasks only: each carries its window's size)
    ra.spec_num = 8;
    ra.spec_extra = kSpecExtraLevels;
    ra.cap_inkernel = (int32_t)std::max<int64_t>(kInKernelBeamCap, qp->beam_width);
    ra.sub_base0 = ra.sub_cap = (int32_t)(nq * maxt);
    ra.sub_count = W.ints.p + I_SUB_COUNT;
    ra.big_list = W.list_big.p;
    ra.big_count = W.ints.p + I_BIG_COUNT;
    ra.big_stride = W.big_stride;
    ra.ctr = W.ctr.p;
    if (launch_route(ra, st)) throw HipError(std::string("k_route: ") + launch_last_error());
    CostArgs ca{};
    ca.tasks = W.tasks.p;
    ca.qtask_cnt = W.qtask_cnt.p;
    ca.parts = I->view.parts;
    ca.nq = nq;
    ca.maxt = maxt;
    ca.k = k;
    ca.beam = ra.beam;
    ca.max_beam = ra.max_beam;
    ca.mult = (int32_t)std::min<int64_t>(std::max<int64_t>(qp->final_beam_multiply, 1), INT32_MAX);
    ca.cost = W.dist_stage.p;
    if (launch_task_cost(ca, st)) throw HipError(std::string("k_task_cost: ") + launch_last_error());
    HIP_CHECK(hipMemcpyAsync(cost, W.dist_stage.p, (size_t)nq * 4, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
  } catch (HipError &e) {
    return fail(WANN_ERR_HIP, e.what());
  } catch (std::exception &e) {
    return fail(WANN_ERR_INVALID, e.what());
  }
  return WANN_OK;
}

namespace {
// one replica's share of a host-buffer call: stage, search (queries keep their global numbers), copy back
void search_host_one(wann_index &T, const void *queries, const float *ranges, int64_t nq, int64_t qid_base, const char *method,
                     const wann_query_params &qp, uint32_t *ids, float *dists) {
  std::lock_guard<std::mutex> lk(T.mu);
  HIP_CHECK(hipSetDevice(T.device));
  const Tuning tune = snapshot_tuning(T);  // (WANN_TEST_HOOKS=1 only: re-read)
  Workspace &W = T.ws;
  const int64_t d = T.host().spec.d;
  if (qp.k <= 0 || qp.k > 1024) throw std::runtime_error("k must be in [1, 1024]");
  W.q_stage.ensure((size_t)nq * d);
  W.r_stage.ensure((size_t)nq * 2);
  W.id_stage.ensure((size_t)nq * qp.k);
  W.dist_stage.ensure((size_t)nq * qp.k);
  hipStream_t st = T.own_stream;
  std::vector<float> qf;  // host queries arrive in the index's element type
  if (nq && T.dtype != WANN_DTYPE_F32) {
    qf = bytes_to_float(T.dtype, queries, nq * d);
    queries = qf.data();
  }
  if (nq) {
    HIP_CHECK(hipMemcpyAsync(W.q_stage.p, queries, (size_t)nq * d * 4, hipMemcpyHostToDevice, st));
    HIP_CHECK(hipMemcpyAsync(W.r_stage.p, ranges, (size_t)nq * 8, hipMemcpyHostToDevice, st));
  }
  run_batch(T, W, T.side_stream, T.last, W.q_stage.p, W.r_stage.p, nq, qid_base, method, qp, W.id_stage.p, W.dist_stage.p, st, tune);
  if (nq) {
    HIP_CHECK(hipMemcpyAsync(ids, W.id_stage.p, (size_t)nq * qp.k * 4, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipMemcpyAsync(dists, W.dist_stage.p, (size_t)nq * qp.k * 4, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
  }
}
}  // namespace

int wann_batch_search(wann_index *I, const void *queries, const float *ranges, int64_t nq, const char *method,
                      const wann_query_params *qp, uint32_t *ids, float *dists) {
  if (!I || !qp || nq < 0 || (nq > 0 && (!queries || !ranges || !ids || !dists)))
    return fail(WANN_ERR_INVALID, "invalid argument to wann_batch_search");
  const int G = 1 + (int)I->replicas.size();
  if (G == 1 || nq < G) {
    try {
      search_host_one(*I, queries, ranges, nq, 0, method, *qp, ids, dists);
    } catch (HipError &e) {
      return fail(WANN_ERR_HIP, e.what());
    } catch (std::exception &e) {
      return fail(WANN_ERR_INVALID, e.what());
    }
    return WANN_OK;
  }
  // In-process multi-device mode (WANN_DEVICES): contiguous shards that keep their global query numbers (the reference uses
  // a query's row number as its own id, range_filter_tree.h:62-96 + beamSearch.h:128), one host thread and one stream per
  // replica, rows land in the caller's arrays.
  const int64_t d = I->H.spec.d, esz = I->dtype == WANN_DTYPE_F32 ? 4 : 1;
  std::vector<std::thread> threads;
  std::vector<int> codes((size_t)G, WANN_OK);
  std::vector<std::string> errs((size_t)G);
  for (int g = 0; g < G; g++) {
    const int64_t base = nq / G, rem = nq % G;
    const int64_t lo = g * base + std::min<int64_t>(g, rem), cnt = base + (g < rem ? 1 : 0);
    wann_index *T = g == 0 ? I : I->replicas[(size_t)g - 1].get();
    threads.emplace_back([=, &codes, &errs] {
      try {
        search_host_one(*T, (const char *)queries + lo * d * esz, ranges + 2 * lo, cnt, lo, method, *qp, ids + lo * qp->k, dists + lo * qp->k);
      } catch (HipError &e) {
        codes[(size_t)g] = WANN_ERR_HIP;
        errs[(size_t)g] = e.what();
      } catch (std::exception &e) {
        codes[(size_t)g] = WANN_ERR_INVALID;
        errs[(size_t)g] = e.what();
      }
    });
  }
  for (auto &t : threads) t.join();
  (void)hipSetDevice(I->device);
  for (int g = 0; g < G; g++)
    if (codes[(size_t)g] != WANN_OK) return fail(codes[(size_t)g], "replica " + std::to_string(g) + ": " + errs[(size_t)g]);
  // counters of the call: work summed over the replicas, times of the slowest one
  wann_counters sum = I->last;
  for (auto &R : I->replicas) {
    const wann_counters &c = R->last;
    sum.beam_searches += c.beam_searches;
    sum.hops += c.hops;
    sum.dist_cmps += c.dist_cmps;
    sum.brute_rows += c.brute_rows;
    sum.label_reads += c.label_reads;
    sum.rounds = std::max(sum.rounds, c.rounds);
    sum.spec_searches += c.spec_searches;
    sum.spec_hops += c.spec_hops;
    sum.spec_dist_cmps += c.spec_dist_cmps;
    sum.gemm_queries += c.gemm_queries;
    sum.gemm_unproven += c.gemm_unproven;
    sum.gemm_rescued += c.gemm_rescued;
    sum.recovered_continuations += c.recovered_continuations;
    sum.deep_handoffs += c.deep_handoffs;
    sum.lookaheads_used += c.lookaheads_used;
    sum.lookaheads_issued += c.lookaheads_issued;
    sum.big_searches += c.big_searches;
    sum.big_hops += c.big_hops;
    sum.packet_hops += c.packet_hops;
    sum.own_scorings += c.own_scorings;
    sum.prefetched_hops += c.prefetched_hops;
    sum.poll_timeouts += c.poll_timeouts;
    sum.device_ms = std::max(sum.device_ms, c.device_ms);
    sum.search_kernel_ms = std::max(sum.search_kernel_ms, c.search_kernel_ms);
  }
  I->last = sum;
  return WANN_OK;
}

int wann_get_counters(const wann_index *I, wann_counters *out) {
  if (!I || !out) return fail(WANN_ERR_INVALID, "null argument");
  *out = I->last;
  return WANN_OK;
}

int64_t wann_num_points(const wann_index *I) { return I ? I->H.spec.n : -1; }
int64_t wann_dim(const wann_index *I) { return I ? I->H.spec.d : -1; }
int64_t wann_num_levels(const wann_index *I) { return I ? (int64_t)I->H.levels.size() : -1; }
int64_t wann_level_size(const wann_index *I, int64_t level) {
  if (!I || level < 0 || level >= (int64_t)I->H.levels.size()) return -1;
  return (int64_t)I->H.levels[level].size();
}
int wann_partition_range(const wann_index *I, int64_t level, int64_t idx, int64_t *start, int64_t *end) {
  if (!I || level < 0 || level >= (int64_t)I->H.levels.size() || idx < 0 || idx >= (int64_t)I->H.levels[level].size())
    return fail(WANN_ERR_INVALID, "partition out of range");
  const HostPart &P = I->H.levels[level][idx];
  *start = P.start;
  *end = P.start + P.n;
  return WANN_OK;
}
int wann_partition_graph(const wann_index *I, int64_t level, int64_t idx, int32_t *rows, int64_t cap_rows, int64_t max_degree) {
  if (!I || !rows || level < 0 || level >= (int64_t)I->H.levels.size() || idx < 0 || idx >= (int64_t)I->H.levels[level].size())
    return fail(WANN_ERR_INVALID, "partition out of range");
  const HostPart &P = I->H.levels[level][idx];
  if (P.g.n != P.n || cap_rows < P.n) return fail(WANN_ERR_INVALID, "no graph / buffer too small");
  if (max_degree != (int64_t)P.g.maxdeg)
    return fail(WANN_ERR_INVALID, "max_degree " + std::to_string((long long)max_degree) + " does not match the index's R = " +
                                      std::to_string((long long)P.g.maxdeg) + " (rows are R+1 ints wide)");
  memcpy(rows, P.g.rows.data(), (size_t)P.n * (size_t)(P.g.maxdeg + 1) * 4);
  return WANN_OK;
}
int64_t wann_max_degree(const wann_index *I) { return I ? I->H.spec.R : -1; }
int64_t wann_device_bytes(const wann_index *I) { return I ? I->device_bytes : -1; }
int wann_num_replicas(const wann_index *I) { return I ? 1 + (int)I->replicas.size() : -1; }

int wann_build_cache_shard(int kind, int metric, int dtype, const void *points, int64_t n, int64_t d,
                           const float *labels, int32_t cutoff, double split_factor, double shift_factor,
                           const wann_build_params *bp, int shard, int nshards, int build_threads) {
  if (dtype != WANN_DTYPE_F32 && dtype != WANN_DTYPE_U8 && dtype != WANN_DTYPE_I8) return fail(WANN_ERR_INVALID, "unknown dtype");
  if (!bp || !bp->cache_path || !*bp->cache_path) return fail(WANN_ERR_INVALID, "cache_path required");
  if (nshards <= 0 || shard < 0 || shard >= nshards) return fail(WANN_ERR_INVALID, "bad shard");
  try {
    HostIndex H;
    H.spec = make_spec(kind, metric, dtype, n, d, cutoff, split_factor, shift_factor, bp, build_threads);
    build_host_index(H, points, labels, shard, nshards);
  } catch (std::exception &e) {
    return fail(WANN_ERR_INVALID, e.what());
  }
  return WANN_OK;
}

}  // extern "C"
