"""Dev tool: the counter passes of a mid-window-fraction leg (tools/gpu_jobs/r06_first.sh: separate rocprofv3 --pmc runs of
`bench.py --fraction p --setting 80,1`) -> profiles/<tag>_mid_fraction_2pow<p>_pmc_traffic.json, the file bench.py's
measured_traffic() looks up for roofline.traffic / per_fraction[*].traffic.
Usage: python tools/summarize_mid_pmc.py gpurun_out/r06a r06 -9 -6"""
import csv, glob, json, os, statistics, sys

src, tag = sys.argv[1], sys.argv[2]
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def counters(p):
    acc = {}
    for f in glob.glob(os.path.join(src, f"pmc{p}_g*", "**", "*counter_collection.csv.sel.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            kn = r["Kernel_Name"]
            if "k_search" not in kn and "k_brute" not in kn:
                continue
            kind = kn[kn.index("k_search" if "k_search" in kn else "k_brute"):].split("(")[0]
            acc.setdefault(kind, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return acc


def timed(vals):  # the timed launches: warm-up / ground-truth batches of other shapes aside
    top = max(vals)
    v = [x for x in vals if x > 0.5 * top] if top > 0 else list(vals)
    return statistics.median(v), len(v)


for ptok in sys.argv[3:]:  # '-9' (directories pmc-9_g*) or '-8_40,1' (directories pmc-8_40,1_g*)
    p = ptok
    frac = int(ptok.split('_')[0])
    line = None
    for g in range(1, 9):
        fn = os.path.join(src, f"pmc{p}_g{g}.json")
        if os.path.exists(fn):
            ls = [l for l in open(fn).read().splitlines() if l.startswith("{")]
            if ls:
                line = json.loads(ls[-1])
                break
    acc = counters(p)
    per_kernel, total_fetch, total_write = {}, 0.0, 0.0
    for kind, cs in sorted(acc.items()):
        rec = {}
        for name, vals in sorted(cs.items()):
            med, n = timed(vals)
            rec[name] = dict(median=med, launches=n)
        if "FETCH_SIZE" in rec:
            rec["fetched_bytes_per_launch"] = rec["FETCH_SIZE"]["median"] * 1024 * 2
            total_fetch += rec["fetched_bytes_per_launch"]
        if "WRITE_SIZE" in rec:
            rec["WRITE_SIZE_bytes_uncorrected"] = rec["WRITE_SIZE"]["median"] * 1024
            total_write += rec["WRITE_SIZE_bytes_uncorrected"]
        if "TCC_HIT_sum" in rec and "TCC_MISS_sum" in rec:
            h, m = rec["TCC_HIT_sum"]["median"], rec["TCC_MISS_sum"]["median"]
            rec["l2_hit_rate"] = round(h / (h + m), 4)
        if "SQ_WAVE_CYCLES" in rec:
            wc = rec["SQ_WAVE_CYCLES"]["median"]
            for nm in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
                if nm in rec:
                    rec[nm + "_share_of_wave_cycles"] = round(rec[nm]["median"] / wc, 4)
        per_kernel[kind] = rec
    alg = line["roofline"]["algorithmic_bytes_per_step"] if line else None  # (graph rows + scored vectors + labels; exact scans are not in it)
    out = dict(
        what=f"SIFT-1M-like 2-WST, window fraction 2^{frac}, 10 000 queries, the setting named below: rocprofv3 --pmc passes (one counter group per run) of "
             "bench.py --fractions headline --fraction p --setting beam,mult --pipeline 0.  Under --pmc the runtime serialises dispatches: the companion launch "
             "(k_search<0, 1>) runs BEFORE the ordinary one instead of beside it, its pollers give up, and continuations / look-aheads run in "
             "follow-up launches -- the bytes are those of the batch, the kernels' durations are not the concurrent batch's",
        kernel=" + ".join(sorted(per_kernel)) + " (sum per batch)", n=1_000_000, nq=10_000, fraction=frac,
        beam=line["config"]["beam"] if line else 80, mult=line["config"]["final_beam_multiply"] if line else 1,
        scan_only=bool(line and line["roofline"]["hops_per_step"] == 0),  # (every window takes the exact scan: the setting changes nothing)
        correction="FETCH_SIZE (KiB) x 1024 x 2: TCC_EA0_RDREQ_32B is 0 in every pass, i.e. every request is a full line; bytes = RDREQ x 128 B "
                   "(profiles/r04_fetch_size_calibration.json for row gathers; profiles/r06_fetch_size_probe_calibration.json: a single-dword random "
                   "probe is ONE request = one 128-byte line)",
        hbm_bytes_per_launch=int(total_fetch), write_size_bytes_uncorrected=int(total_write),
        algorithmic_bytes_per_launch=alg, fetched_over_algorithmic=round(total_fetch / alg, 3) if alg else None,
        work_per_step=None if not line else {k: line["roofline"][k] for k in ("searches_per_step", "hops_per_step", "dist_cmps_per_step")},
        per_kernel=per_kernel)
    sfx = "" if not line or (line["config"]["beam"], line["config"]["final_beam_multiply"]) == (80, 1) else f"_beam{line['config']['beam']}x{line['config']['final_beam_multiply']}"
    fn = os.path.join(REPO, "profiles", f"{tag}_fraction_2pow{frac}{sfx}_pmc_traffic.json")
    json.dump(out, open(fn, "w"), indent=1)
    print(fn, "fetched/algorithmic", out["fetched_over_algorithmic"], "GB fetched", round(total_fetch / 1e9, 2))
