"""Dev tool: turn a round's profiler outputs (gpurun_out/<dir> of tools/gpu_jobs/rNN_final.sh) into the summaries kept under profiles/:
kernel-stat tables, the FETCH_SIZE passes as *_pmc_traffic.json (what bench.py's roofline.traffic fields look up), MFMA counters.
Usage: python tools/summarize_profiles.py gpurun_out/r05final r05"""
import csv, glob, json, os, shutil, statistics, sys

src, tag = sys.argv[1], sys.argv[2]
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(REPO, "profiles")


def one(pattern):
    g = glob.glob(os.path.join(src, pattern))
    return g[0] if g else None


def kernel_rows(path, match):
    rows = []
    for r in csv.DictReader(open(path)):
        if match in r["Name"]:
            rows.append(dict(name=r["Name"], calls=int(r["Calls"]), avg_ms=float(r["AverageNs"]) / 1e6, min_ms=float(r["MinNs"]) / 1e6, max_ms=float(r["MaxNs"]) / 1e6))
    return rows


def counters(path, match):
    acc = {}
    for r in csv.DictReader(open(path)):
        if match in r["Kernel_Name"]:
            acc.setdefault((r["Kernel_Name"], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    return acc


def copy(pattern, name):
    p = one(pattern)
    if p:
        shutil.copy(p, os.path.join(out, name))
    return p


def line(path):
    try:
        return json.loads([l for l in open(path).read().splitlines() if l.startswith("{")][-1])
    except Exception:
        return None


summary = {}
# headline
copy("head_kt/*/*kernel_stats.csv", f"{tag}_rocprofv3_kernel_stats.csv")
copy("head_kt/*/*domain_stats.csv", f"{tag}_rocprofv3_domain_stats.csv")
hl = line(os.path.join(src, "head_kt.json"))
if hl:
    json.dump(hl, open(os.path.join(out, f"{tag}_rocprofv3_bench_line.json"), "w"))
    summary["headline_under_profiler"] = dict(kernel_ms_per_step=hl["roofline"]["kernel_ms_per_step"], frac=hl["roofline"]["frac"],
                                              kernel_stats=kernel_rows(one("head_kt/*/*kernel_stats.csv"), "k_search"))
p = one("head_fetch/*/*counter_collection.csv.sel.csv")
if p and hl:
    c = counters(p, "k_search<0, 0>")
    vals = [v for (k, n), vs in c.items() if n == "FETCH_SIZE" for v in vs]
    vals = [v for v in vals if v > 0.5 * max(vals)]  # (the timed launches; warm-up / ground-truth batches of other shapes aside)
    med = statistics.median(vals)
    rec = dict(kernel="wann::dt_f32::k_search<0, 0>", n=1000000, nq=10000, fraction=-3, beam=80, mult=1, launches=len(vals), FETCH_SIZE_median=med,
               FETCH_SIZE_unit="KiB as reported by rocprofv3 (TCC_EA0_RDREQ x 64 B / 1024)",
               correction="x2 (profiles/r04_fetch_size_calibration.json: FETCH_SIZE x 2 = known bytes of this kernel's gather pattern within 1.8 %; "
                          "Infinity-Cache hits are counted: fabric-side traffic, not DRAM traffic)",
               hbm_bytes_per_launch=int(med * 1024 * 2), algorithmic_bytes_per_launch=hl["roofline"]["algorithmic_bytes_per_step"],
               command="rocprofv3 --pmc FETCH_SIZE --output-format csv -- python3 bench.py --steps 5 --warmup 1 --fractions headline --configs none "
                       "--no-cpu-baseline --setting 80,1 --pipeline 0",
               note="rotating batches; under --pmc the runtime serialises dispatches: the companion launch's pollers give up and every search runs in k_search<0, 0>")
    json.dump(rec, open(os.path.join(out, f"{tag}_pmc_traffic.json"), "w"), indent=1)
    summary["headline_traffic_ratio"] = rec["hbm_bytes_per_launch"] / rec["algorithmic_bytes_per_launch"]
# mid fractions
for p_ in (-6, -8, -9, -11):
    if copy(f"mid{p_}_kt/*/*kernel_stats.csv", f"{tag}_mid_fraction_2pow{p_}_kernel_stats.csv"):
        l = line(os.path.join(src, f"mid{p_}_kt.json"))
        if l:
            json.dump(l, open(os.path.join(out, f"{tag}_mid_fraction_2pow{p_}_bench_line.json"), "w"))
        summary[f"2^{p_}"] = kernel_rows(one(f"mid{p_}_kt/*/*kernel_stats.csv"), "k_search")
# configs
for cfg, kern, setting in (("glove", "k_search<1, 0>", (40, 1)), ("deep", "k_search<1, 0>", (80, 1))):
    copy(f"{cfg}_kt/*/*kernel_stats.csv", f"{tag}_config_{cfg}_kernel_stats.csv")
    l = line(os.path.join(src, f"{cfg}_kt.json"))
    if l:
        json.dump(l, open(os.path.join(out, f"{tag}_config_{cfg}_bench_line.json"), "w"))
    p = one(f"{cfg}_fetch/*/*counter_collection.csv.sel.csv")
    if p and l:
        shutil.copy(p, os.path.join(out, f"{tag}_config_{cfg}_fetch_size.csv"))
        c = counters(p, kern)
        vals = [v for (k, n), vs in c.items() if n == "FETCH_SIZE" for v in vs]
        vals = [v for v in vals if v > 0.5 * max(vals)]
        med = statistics.median(vals)
        rec = dict(config=cfg, kernel=f"wann::dt_f32::{kern}", beam=setting[0], mult=setting[1], launches=len(vals), FETCH_SIZE_median=med,
                   correction="x2 (see the headline's *_pmc_traffic.json)", hbm_bytes_per_launch=int(med * 1024 * 2),
                   algorithmic_bytes_per_launch=int(l["algorithmic_gb_per_batch"] * 1e9),
                   command=f"rocprofv3 --pmc FETCH_SIZE --output-format csv -- python3 tools/bench_configs.py --config {cfg} --threads '' --setting {setting[0]},{setting[1]}")
        json.dump(rec, open(os.path.join(out, f"{tag}_config_{cfg}_pmc_traffic.json"), "w"), indent=1)
        summary[f"{cfg}_traffic_ratio"] = rec["hbm_bytes_per_launch"] / rec["algorithmic_bytes_per_launch"]
        summary[f"{cfg}_kernel"] = kernel_rows(one(f"{cfg}_kt/*/*kernel_stats.csv"), "k_search")
# prefilter
copy("prefilter_kt/*/*kernel_stats.csv", f"{tag}_prefilter_rocprofv3_kernel_stats.csv")
for nm in ("prefilter", "prefilter_d512"):
    l = line(os.path.join(src, nm + ".json"))
    if l:
        json.dump(l, open(os.path.join(out, f"{tag}_{'prefilter_config5' if nm == 'prefilter' else nm}.json"), "w"))
pm, pf, pk = one("prefilter_mfma/*/*counter_collection.csv.sel.csv"), one("prefilter_fetch/*/*counter_collection.csv.sel.csv"), one("prefilter_kt/*/*kernel_stats.csv")
if pm and pf and pk:
    cm, cf = counters(pm, "k_gemm_scores"), counters(pf, "k_gemm_scores")
    mx = lambda c, n: max(v for (k, nn), vs in c.items() if nn == n for v in vs)
    kr = kernel_rows(pk, "k_gemm_scores")
    mops, busy, sqb, fetch = mx(cm, "SQ_INSTS_VALU_MFMA_MOPS_BF16"), mx(cm, "SQ_VALU_MFMA_BUSY_CYCLES"), mx(cm, "SQ_BUSY_CYCLES"), mx(cf, "FETCH_SIZE")
    kms = max(r["max_ms"] for r in kr)
    gflop = mops * 512 / 1e9
    rec = dict(kernel="k_gemm_scores<112> on the adversarial batch (9 900 queries, 100 window groups of 10 000 points, d = 100); the launches with work are the maxima",
               SQ_INSTS_VALU_MFMA_MOPS_BF16_max=mops, executed_gflop=round(gflop, 2), SQ_VALU_MFMA_BUSY_CYCLES_max=busy, SQ_BUSY_CYCLES_max=sqb,
               kernel_ms_max_from_kernel_trace=round(kms, 4), executed_tflops=round(gflop / kms, 1), frac_of_dense_bf16_peak_2500=round(gflop / kms / 2500, 3),
               FETCH_SIZE_KiB_max=fetch, fetched_mb_x2=round(fetch * 1024 * 2 / 1e6, 1), algorithmic_mb=448.0, fetched_over_algorithmic=round(fetch * 1024 * 2 / 448e6, 3),
               hbm_frac_of_8TBs=round(448e6 / (kms * 1e-3) / 8e12, 3), kernel_stats=kr)
    json.dump(rec, open(os.path.join(out, f"{tag}_prefilter_mfma_counters.json"), "w"), indent=1)
    json.dump(dict(config="adverse", kernel="k_gemm_scores<112>", hbm_bytes_per_launch=int(fetch * 1024 * 2), algorithmic_bytes_per_launch=448000000,
                   command="rocprofv3 --pmc FETCH_SIZE --output-format csv -- python3 tools/bench_prefilter.py"),
              open(os.path.join(out, f"{tag}_config_adverse_pmc_traffic.json"), "w"), indent=1)
    summary["prefilter"] = rec
fw = {c: line(os.path.join(src, c + ".json")) for c in ("fenwick", "three_split")}
if all(fw.values()):
    for v in fw.values():
        v.pop("sweep", None)
    json.dump(fw, open(os.path.join(out, f"{tag}_fenwick_three_split_1m.json"), "w"))
print(json.dumps(summary, indent=1)[:6000])
