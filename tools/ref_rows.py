"""Child process of tests/test_gpu_fullsize.py: the REAL reference (oracle/_ref build, g++ straight on /root/reference's
python_bindings.cpp) on one of BASELINE.json's full-size configurations -- index loaded from the graph files the GPU builder left
in the cache directory, same data laws and seeds as bench.py / tools/bench_configs.py / tools/bench_prefilter.py -- answering
every leg of an .npz (W|name windows, set|name (beam, mult)) and writing its rows to another .npz (ids|name, dists|name).
The comparison itself happens in the test.  Exit code 3 = no reference build present.

  python tools/ref_rows.py --config sift|glove|deep|adverse --cache DIR --legs in.npz --out out.npz [--threads T]"""
import argparse, os, sys, time

ap = argparse.ArgumentParser()
ap.add_argument("--config", required=True)
ap.add_argument("--cache", required=True)
ap.add_argument("--legs", required=True)
ap.add_argument("--out", required=True)
ap.add_argument("--threads", type=int, default=0)
args = ap.parse_args()
os.environ["PARLAY_NUM_THREADS"] = str(args.threads or min(32, os.cpu_count() or 1))
os.environ["WANN_NO_TORCH"] = "1"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
from oracle import oracle as orc
from util import quiet_stdout
import fullsize_configs as fc

ref = orc.load_reference(prefer=("x86-64-v4", "native", "x86-64-v3"))
if ref is None:
    print("no reference build under oracle/_ref", file=sys.stderr)
    sys.exit(3)
cfg = fc.CONFIGS[args.config]
X, Q, labels = fc.make_data(args.config)
t0 = time.time()
with quiet_stdout():
    idx = fc.make_index(ref, args.config, X, labels, args.cache)
print(f"[ref_rows] {args.config}: reference index ready in {time.time() - t0:.1f}s", file=sys.stderr, flush=True)
legs = np.load(args.legs)
out = {}
for name in sorted({k.split("|", 1)[1] for k in legs.files if k.startswith("W|")}):
    beam, mult = (int(x) for x in legs["set|" + name])
    W = legs["W|" + name].astype(np.float64)
    a = (Q, W, Q.shape[0]) + ((cfg["method"],) if cfg["method"] is not None else ())
    t0 = time.time()
    with quiet_stdout():
        ids, dists = idx.batch_search(*a, fc.query_params(ref, beam, mult))
    print(f"[ref_rows]   {name}: {Q.shape[0] / (time.time() - t0):,.0f} QPS", file=sys.stderr, flush=True)
    out["ids|" + name] = ids
    out["dists|" + name] = dists
np.savez(args.out, **out)
