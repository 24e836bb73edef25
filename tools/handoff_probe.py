"""Dev tool: does the companion launch's deep-chain pollers get their CUs?  Runs the deep-10M-like (or SIFT-1M-like) batch repeatedly and
prints, per call, the device time and how many chains were handed to pollers (deep_handoffs) -- 0 on a call means the ordinary
launch had booked every CU before the pollers were placed.  Usage: python tools/handoff_probe.py deep|sift [calls]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import window_ann as wa
import fullsize_configs as fc
name = sys.argv[1] if len(sys.argv) > 1 else "deep"
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 24
cfg = fc.CONFIGS[name]
X, Q, labels = fc.make_data(name)
cache = f"/tmp/wann_fullsize_cache/{name}_n{cfg['n']}/"
t0 = time.time()
idx = fc.make_index(wa, name, X, labels, cache)
print(f"index ready in {time.time() - t0:.0f}s", flush=True)
beam, frac = (40, -6) if name == "glove" else (80, -3)
W = fc.fraction_windows(labels, cfg["nq"], frac, 1997).astype(np.float32)
a = (Q, W, cfg["nq"]) + ((cfg["method"],) if cfg["method"] is not None else ())
hist = []
for i in range(calls):
    idx.batch_search(*a, fc.query_params(wa, beam, 1))
    c = idx.counters()
    hist.append((c["deep_handoffs"], round(c["device_ms"], 2), round(c.get("search_kernel_ms", 0.0), 2)))
print("per call (deep_handoffs, device_ms, search_kernel_ms):", hist)
