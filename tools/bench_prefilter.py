"""BASELINE.json configs[4]: adversarial-style data (100 clusters x 10 000 points, d = 100, unit norm, MIPS,
one window per cluster = 1 % of the points, 9 900 queries), PrefilterIndex brute force:
dense MFMA path vs the exact per-query scan vs the real reference.  Run from the repo root."""
import json, os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
os.environ.setdefault("PARLAY_NUM_THREADS", str(os.cpu_count()))
import torch
import window_ann as wa
from oracle import oracle as orc
from util import quiet_stdout

rng = np.random.default_rng(0)
nclu, per, d = 100, 10000, 100
n = nclu * per
cent = rng.standard_normal((nclu, d)).astype(np.float32)
X = cent[np.repeat(np.arange(nclu), per)] + 0.1 * rng.standard_normal((n, d)).astype(np.float32)
X /= np.linalg.norm(X, axis=1, keepdims=True)
labels = (np.repeat(np.arange(nclu), per) - 0.5 + rng.random(n)).astype(np.float32)
qc = np.repeat(np.arange(nclu), 99)
Q = cent[(qc + 1 + rng.integers(0, nclu - 1, qc.size)) % nclu] + 0.1 * rng.standard_normal((qc.size, d)).astype(np.float32)
Q /= np.linalg.norm(Q, axis=1, keepdims=True)
Q = Q.astype(np.float32)
W = np.stack([qc - 0.5, qc + 0.5], 1).astype(np.float32)
nq = Q.shape[0]
idx = wa.PrefilterIndexFloatMips(X, labels)
qp = wa.QueryParams(10, 10, 1.35, 10**7, 10**4, 1, 10000, None, False)
dev = torch.device("cuda:0")
Qt, Wt = torch.from_numpy(Q).to(dev), torch.from_numpy(W).to(dev)
it, dt = torch.empty((nq, 10), dtype=torch.int32, device=dev), torch.empty((nq, 10), dtype=torch.float32, device=dev)
out = {}
for name, env in (("mfma", None), ("scan", "1")):
    if env: os.environ["WANN_NO_GEMM"] = env
    else: os.environ.pop("WANN_NO_GEMM", None)
    for _ in range(2):
        idx.batch_search_device(Qt.data_ptr(), Wt.data_ptr(), nq, 0, "", qp, it.data_ptr(), dt.data_ptr(), 0)
    t = time.perf_counter()
    reps = 5
    for _ in range(reps):
        idx.batch_search_device(Qt.data_ptr(), Wt.data_ptr(), nq, 0, "", qp, it.data_ptr(), dt.data_ptr(), 0)
    ms = (time.perf_counter() - t) / reps * 1e3
    out[name] = dict(ms=round(ms, 3), qps=round(nq / ms * 1e3), counters=idx.counters(), ids=it.cpu().numpy().view(np.uint32).copy(), d=dt.cpu().numpy().copy())
same = np.array_equal(out["mfma"]["d"], out["scan"]["d"])
ref = orc.load_reference(prefer=("x86-64-v4", "native"))
cpu = None
if ref is not None:
    with quiet_stdout():
        ridx = ref.PrefilterIndexFloatMips(X, labels)
    best = 1e9
    for _ in range(3):
        t = time.perf_counter()
        with quiet_stdout():
            rids, rd = ridx.batch_search(Q, W.astype(np.float64), nq, ref.QueryParams(10, 10, 1.35, 10**7, 10**4, 1, 10000, None, False))
        best = min(best, time.perf_counter() - t)
    cpu = dict(qps=round(nq / best), threads=int(os.environ["PARLAY_NUM_THREADS"]), dists_identical=bool(np.array_equal(rd, out["mfma"]["d"])))
flops = 2.0 * nq * per * d
print(json.dumps(dict(workload="adversarial 100x10000 d=100 MIPS, 9900 queries, window = 1 cluster", mfma_ms=out["mfma"]["ms"], mfma_qps=out["mfma"]["qps"],
                      scan_ms=out["scan"]["ms"], scan_qps=out["scan"]["qps"], mfma_equals_scan=bool(same),
                      gemm_queries=out["mfma"]["counters"]["gemm_queries"], gemm_unproven=out["mfma"]["counters"]["gemm_unproven"], gemm_rescued=out["mfma"]["counters"]["gemm_rescued"], device_ms=round(out["mfma"]["counters"]["device_ms"], 3), gemm_tflops_incl_select=round(flops / out["mfma"]["ms"] / 1e9, 2), cpu_reference=cpu)))
