"""BASELINE.json configs[4]: adversarial-style data (100 clusters x 10 000 points, d = 100, unit norm, MIPS, 9 900 cross-cluster
queries; experiments/generate_advserial_dataset.py:8-69), PrefilterIndex brute force (src/prefiltering.h:154-204):
  (i)  native windows [c - 0.5, c + 0.5] = one cluster = 1 % of the points: dense MFMA path vs the exact per-query scan
  (ii) synthetic 2^-12 windows (244 points): the exact scan kernel k_brute against its HBM roofline
each against the REAL reference at its best thread count (child processes: the reference fixes its thread count at first use).
Run from the repo root.  Prints one JSON object."""
import json, os, subprocess, sys, time
os.environ.setdefault("WANN_TEST_HOOKS", "1")  # this tool flips WANN_* switches between calls on one index
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))


def make():
    rng = np.random.default_rng(0)
    nclu, per, d = 100, 10000, int(os.environ.get("WANN_PF_DIM", "100"))  # (WANN_PF_DIM=512: the RedCaps row length)
    n = nclu * per
    cent = rng.standard_normal((nclu, d)).astype(np.float32)
    X = cent[np.repeat(np.arange(nclu), per)] + 0.1 * rng.standard_normal((n, d)).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    labels = (np.repeat(np.arange(nclu), per) - 0.5 + rng.random(n)).astype(np.float32)
    qc = np.repeat(np.arange(nclu), 99)
    Q = cent[(qc + 1 + rng.integers(0, nclu - 1, qc.size)) % nclu] + 0.1 * rng.standard_normal((qc.size, d)).astype(np.float32)
    Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    Q = Q.astype(np.float32)
    W1 = np.stack([qc - 0.5, qc + 0.5], 1).astype(np.float32)
    # (ii) 2^-12 of the points (244) at a random place of the label order: [label[s], label[s + 244]]
    ls = np.sort(labels)
    w = int(n * 2.0 ** -12)
    st = np.random.default_rng(5).integers(1, n - w - 1, size=Q.shape[0])
    W2 = np.stack([ls[st], ls[st + w]], 1).astype(np.float32)
    return X, Q, labels, W1, W2, per, w


if len(sys.argv) > 2 and sys.argv[1] == "--ref-worker":  # child: the real reference with PARLAY_NUM_THREADS from the environment
    os.environ["WANN_NO_TORCH"] = "1"
    from oracle import oracle as orc
    from util import quiet_stdout
    X, Q, labels, W1, W2, per, w = make()
    res = np.load(sys.argv[2])
    ref = orc.load_reference(prefer=("x86-64-v4", "native"))
    assert ref is not None
    with quiet_stdout():
        ridx = ref.PrefilterIndexFloatMips(X, labels)
    out = {}
    for name, W in (("native", W1), ("p12", W2)):
        best = 1e9
        for _ in range(3):
            t = time.perf_counter()
            with quiet_stdout():
                rids, rd = ridx.batch_search(Q, W.astype(np.float64), Q.shape[0], ref.QueryParams(10, 10, 1.35, 10**7, 10**4, 1, 10000, None, False))
            best = min(best, time.perf_counter() - t)
        out[name] = dict(qps=round(Q.shape[0] / best), dists_identical=bool(np.array_equal(rd, res["d_" + name])))
    print(json.dumps(out))
    sys.exit(0)

os.environ.setdefault("PARLAY_NUM_THREADS", str(os.cpu_count()))
import torch
import window_ann as wa

X, Q, labels, W1, W2, per, w = make()
d = X.shape[1]
nq = Q.shape[0]
idx = wa.PrefilterIndexFloatMips(X, labels)
qp = wa.QueryParams(10, 10, 1.35, 10**7, 10**4, 1, 10000, None, False)
dev = torch.device("cuda:0")
Qt = torch.from_numpy(Q).to(dev)
it, dt = torch.empty((nq, 10), dtype=torch.int32, device=dev), torch.empty((nq, 10), dtype=torch.float32, device=dev)


def timed(Wt, env):
    if env: os.environ["WANN_NO_GEMM"] = env
    else: os.environ.pop("WANN_NO_GEMM", None)
    for _ in range(2):
        idx.batch_search_device(Qt.data_ptr(), Wt.data_ptr(), nq, 0, "", qp, it.data_ptr(), dt.data_ptr(), 0)
    t = time.perf_counter()
    reps = 5
    for _ in range(reps):
        idx.batch_search_device(Qt.data_ptr(), Wt.data_ptr(), nq, 0, "", qp, it.data_ptr(), dt.data_ptr(), 0)
    ms = (time.perf_counter() - t) / reps * 1e3
    return dict(ms=round(ms, 3), qps=round(nq / ms * 1e3), counters=idx.counters(), ids=it.cpu().numpy().view(np.uint32).copy(), d=dt.cpu().numpy().copy())


W1t, W2t = torch.from_numpy(W1).to(dev), torch.from_numpy(W2).to(dev)
if os.environ.get("WANN_PF_ONLY") == "p12":  # dev runs under a profiler: the synthetic windows alone
    r = timed(W2t, None)
    print(json.dumps(dict(ms=r["ms"], device_ms=r["counters"]["device_ms"], brute_rows=int(r["counters"]["brute_rows"]))))
    sys.exit(0)
out = {"mfma": timed(W1t, None), "scan": timed(W1t, "1"), "p12": timed(W2t, None)}
os.environ.pop("WANN_NO_GEMM", None)
same = np.array_equal(out["mfma"]["d"], out["scan"]["d"])
# the reference at several thread counts (its best is what is reported)
res_path = "/tmp/wann_prefilter_gpu_rows.npz"
np.savez(res_path, d_native=out["mfma"]["d"], d_p12=out["p12"]["d"])
cpu = {}
ncpu = os.cpu_count() or 1
for th in ([] if os.environ.get("WANN_PF_NO_REF") else sorted({ncpu, max(1, ncpu // 2), max(1, ncpu // 4), max(1, ncpu // 8)}, reverse=True)):  # (WANN_PF_NO_REF=1: dev runs)
    try:
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--ref-worker", res_path], env=dict(os.environ, PARLAY_NUM_THREADS=str(th)),
                           capture_output=True, text=True, timeout=900)
        r = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
        for name, v in r.items():
            if name not in cpu or v["qps"] > cpu[name]["qps"]:
                cpu[name] = dict(v, threads=th)
    except Exception as e:  # noqa: BLE001
        print(f"[prefilter] reference with {th} threads failed: {e!r}", file=sys.stderr)
flops = 2.0 * nq * per * d
scan_bytes = float(out["p12"]["counters"]["brute_rows"]) * d * 4  # w * d * sizeof(T) per brute-force query
p12_dev_ms = out["p12"]["counters"]["device_ms"]
print(json.dumps(dict(workload=f"adversarial 100x10000 d={d} MIPS, 9900 queries, window = 1 cluster", mfma_ms=out["mfma"]["ms"], mfma_qps=out["mfma"]["qps"],
                      scan_ms=out["scan"]["ms"], scan_qps=out["scan"]["qps"], mfma_equals_scan=bool(same),
                      gemm_queries=out["mfma"]["counters"]["gemm_queries"], gemm_unproven=out["mfma"]["counters"]["gemm_unproven"], gemm_rescued=out["mfma"]["counters"]["gemm_rescued"], device_ms=round(out["mfma"]["counters"]["device_ms"], 3), gemm_tflops_incl_select=round(flops / out["mfma"]["ms"] / 1e9, 2),
                      cpu_reference=cpu.get("native"),
                      # the MFMA leg against its roofline: a window group reads each of its point rows ONCE (padded row of
                      # 16 * ceil(d / 16) floats; 100 groups x 10 000 rows here), so the floor is the HBM's, not the matrix
                      # pipes' (DESIGN.md 3.3b "Round 3"); `achieved` is over the WHOLE call's device time (ten small launches) --
                      # k_gemm_scores alone: profiles/*_prefilter_rocprofv3_kernel_stats.csv
                      roofline=dict(bound="hbm", kernel="k_gemm_scores (whole call: route + grouping + GEMM + select / re-rank + finalize)",
                                    achieved=round(100 * per * (16 * ((d + 15) // 16)) * 4 / (out["mfma"]["counters"]["device_ms"] * 1e-3) / 1e9, 1),
                                    peak=8000.0, unit="GB/s",
                                    frac=round(100 * per * (16 * ((d + 15) // 16)) * 4 / (out["mfma"]["counters"]["device_ms"] * 1e-3) / 1e9 / 8000.0, 4),
                                    algorithmic_gb=round(100 * per * (16 * ((d + 15) // 16)) * 4 / 1e9, 4), traffic=None),
                      synthetic_2pow_minus12=dict(workload=f"same points, synthetic windows of {w} points (2^-12), exact scan (k_brute)", ms=out["p12"]["ms"], qps=out["p12"]["qps"],
                                                  device_ms=round(p12_dev_ms, 4), brute_rows=int(out["p12"]["counters"]["brute_rows"]),
                                                  algorithmic_gb=round(scan_bytes / 1e9, 4),
                                                  roofline=dict(bound="hbm", kernel="k_brute (whole call: route + scan + finalize)", achieved=round(scan_bytes / (p12_dev_ms * 1e-3) / 1e9, 1),
                                                                peak=8000.0, unit="GB/s", frac=round(scan_bytes / (p12_dev_ms * 1e-3) / 1e9 / 8000.0, 4), traffic=None),
                                                  cpu_reference=cpu.get("p12")))))
