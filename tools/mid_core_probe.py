"""Dev tool: the four-wave kernel's general core (beams 129 .. 1 280) on a stand-alone graph -- kernel time per (beam, nq) and
the time per hop and wave slot.  Usage: python tools/mid_core_probe.py [n] [beams] [nqs] [metric d]
(same-box A/B of two builds: LD_LIBRARY_PATH=<dir with the other libwann.so> python tools/mid_core_probe.py ...)"""
import os, re, sys, time, subprocess, tempfile
import numpy as np
os.environ.setdefault("WANN_TEST_HOOKS", "1")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from util import sift_like, unit_mixture
import window_ann as wa
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
BEAMS = tuple(int(x) for x in sys.argv[2].split(',')) if len(sys.argv) > 2 else (160, 320, 640, 1280)
NQS = tuple(int(x) for x in sys.argv[3].split(',')) if len(sys.argv) > 3 else (64, 8192)
metric = int(sys.argv[4]) if len(sys.argv) > 4 else 0
d = int(sys.argv[5]) if len(sys.argv) > 5 else (128 if metric == 0 else 96)
g = (sift_like if metric == 0 else unit_mixture)(n, d, 1234); X = g(n)
cache = f"/tmp/mid_probe_cache_{n}_{metric}_{d}/"; os.makedirs(cache, exist_ok=True)
lab = np.arange(n, dtype=np.float32)
t0 = time.time()
cls = wa.PostfilterVamanaIndexFloatEuclidian if metric == 0 else wa.PostfilterVamanaIndexFloatMips
idx = cls(X, filters=lab, build_params=wa.BuildParams(64, 500, 1.0, cache))
rows = idx.partition_graph(0, 0, 64)
print(f"# graph n={n} d={d} metric={metric} in {time.time()-t0:.0f}s", flush=True)
os.environ["WANN_VERBOSE"] = "1"
for nq in NQS:
    Q = g(nq); qids = np.arange(nq, dtype=np.int64) + 10**7
    for beam in BEAMS:
        # the kernel time is printed to stderr by the library ([wann raw] ...): run the call with stderr captured
        r, w = os.pipe(); saved = os.dup(2); os.dup2(w, 2)
        try:
            for _ in range(2):
                ids, dists, sizes, hops, cmps = wa.raw_beam_search(metric, X, rows, 0, Q, qids, beam)
        finally:
            os.dup2(saved, 2); os.close(w)
        txt = os.read(r, 1 << 16).decode(); os.close(r)
        for ln in txt.splitlines():
            if "phases" in ln: print("   " + ln, flush=True)
        ms = [float(m) for m in re.findall(r": ([0-9.]+) ms", txt)]
        blocks = [int(m) for m in re.findall(r"blocks (\d+)", txt)]
        kind = re.findall(r"kernel kind (\d)", txt)
        slots = blocks[-1] * (4 if kind and kind[-1] == "0" else 1)
        t = min(ms)
        per_hop = t * 1e3 * min(slots, nq) / float(hops.sum())
        print(f"nq={nq:6d} beam={beam:5d} kind {kind[-1]} slots {slots}: kernel {t:8.3f} ms  hops/search {hops.mean():7.1f} cmps/search {cmps.mean():8.0f}  -> {per_hop:5.2f} us per hop and wave", flush=True)
