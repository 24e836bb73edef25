"""Dev tool: FETCH_SIZE pass of a tools/bench_configs.py leg (rocprofv3 --pmc FETCH_SIZE -d <dir>/<name>_fetch -- python3
tools/bench_configs.py --config <name> --setting b,m > <dir>/<name>_fetch.json) -> profiles/<tag>_config_<name>_pmc_traffic.json,
what bench.py's config_traffic() looks up.  Usage: python tools/summarize_config_pmc.py gpurun_out/r06d r06 fenwick three_split sift_u8"""
import csv, glob, json, os, statistics, sys

src, tag = sys.argv[1], sys.argv[2]
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for name in sys.argv[3:]:
    acc = {}
    for f in glob.glob(os.path.join(src, f"{name}_fetch", "**", "*counter_collection.csv.sel.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            kn = r["Kernel_Name"]
            kind = kn[kn.index("k_search" if "k_search" in kn else "k_brute"):].split("(")[0]
            acc.setdefault(kind, []).append(float(r["Counter_Value"]))
    line = json.loads([l for l in open(os.path.join(src, f"{name}_fetch.json")) if l.startswith("{")][-1])
    per, total = {}, 0.0
    for kind, vals in sorted(acc.items()):
        v = [x for x in vals if x > 0.5 * max(vals)] if max(vals) > 0 else vals
        b = statistics.median(v) * 1024 * 2
        per[kind] = dict(FETCH_SIZE_KiB_median=statistics.median(v), launches=len(v), fetched_bytes_per_launch=int(b))
        total += b
    alg = line["algorithmic_gb_per_batch"] * 1e9
    scan = (line.get("scan_gb_per_batch") or 0.0) * 1e9
    out = dict(what=f"tools/bench_configs.py --config {name} under rocprofv3 --pmc FETCH_SIZE (dispatches serialised: k_brute and k_search run one after "
                    "the other here, beside each other in the timed run; the bytes are the batch's)",
               workload=line["workload"], kernel=" + ".join(sorted(per)) + " (sum per batch)", beam=line["setting"]["beam"], mult=line["setting"]["mult"],
               correction="FETCH_SIZE (KiB) x 1024 x 2 (profiles/r04_fetch_size_calibration.json)", per_kernel=per, hbm_bytes_per_launch=int(total),
               algorithmic_bytes_per_launch=int(alg + scan), k_search_algorithmic_bytes=int(alg), k_brute_algorithmic_bytes=int(scan),
               fetched_over_algorithmic=round(total / (alg + scan), 3))
    fn = os.path.join(REPO, "profiles", f"{tag}_config_{name}_pmc_traffic.json")
    json.dump(out, open(fn, "w"), indent=1)
    print(fn, out["fetched_over_algorithmic"], round(total / 1e9, 2), "GB")
