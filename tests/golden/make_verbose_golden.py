"""Generate tests/golden/verbose_golden.json from the REAL reference: the stdout of batch_search with QueryParams.verbose = True
(postfilter_vamana.h:155-185,230), one thread so that the queries' dumps come in query order.  Only the doubling-loop lines are
kept (the tree classes print bucket-search diagnostics and timings around them).  Usage: python tests/golden/make_verbose_golden.py"""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
KEEP = ("Starting optimized postfiltering", "Unfiltered return", "Finished a double", "Final frontier size")
CASES = {
    # name: (class, constructor keywords, method or None, (beam, mult, max_beam), window fraction exponent)
    "postfilter": ("PostfilterVamanaIndexFloatEuclidian", {}, None, (10, 2, 10000), -5),
    "postfilter_maxbeam": ("PostfilterVamanaIndexFloatEuclidian", {}, None, (8, 3, 40), -7),
    "tree": ("VamanaRangeFilterTreeIndexFloatEuclidian", dict(cutoff=300, split_factor=2), "optimized_postfilter", (10, 2, 10000), -4),
}
N, D, NQ, R, L = 2500, 24, 12, 16, 32

WORKER = r'''
import os, sys, json
sys.path.insert(0, os.path.join(%(repo)r, "tests")); sys.path.insert(0, %(repo)r)
import numpy as np
from util import sift_like, distinct_labels, windows
from oracle import oracle as orc
ref = orc.load_reference(prefer=("native", "x86-64-v4"))
cls, kw, method, (beam, mult, maxb), p = %(case)r
g = sift_like(%(n)d, %(d)d, 91)
X, Q = g(%(n)d), g(%(nq)d)
labels = distinct_labels(%(n)d, 92)
W = windows(labels, %(nq)d, p, 93)
labkw = "filters" if cls.startswith("Postfilter") else "filter_values"
idx = getattr(ref, cls)(X, **{labkw: labels}, build_params=ref.BuildParams(%(r)d, %(l)d, 1.0, ""), **kw)
sys.stdout.flush()
print("=====BEGIN", flush=True)
a = (Q, W, %(nq)d) + ((method,) if method else ())
idx.batch_search(*a, ref.QueryParams(10, beam, 1.35, 10**7, 10**4, mult, maxb, None, True))
'''

if __name__ == "__main__":
    repo = os.path.dirname(os.path.dirname(HERE))
    out = {"inputs": dict(n=N, d=D, nq=NQ, R=R, L=L, seeds=[91, 92, 93]), "cases": {}}
    for name, case in CASES.items():
        code = WORKER % dict(repo=repo, case=case, n=N, d=D, nq=NQ, r=R, l=L)
        p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PARLAY_NUM_THREADS="1", WANN_NO_TORCH="1"), capture_output=True, text=True)
        assert p.returncode == 0, p.stderr[-2000:]
        body = p.stdout.split("=====BEGIN", 1)[1]
        lines = [l.strip() for l in body.splitlines() if l.strip().startswith(KEEP)]
        assert lines, name
        out["cases"][name] = dict(cls=case[0], kw=case[1], method=case[2], beam=case[3][0], mult=case[3][1], max_beam=case[3][2], fraction=case[4], lines=lines)
        print(name, len(lines), "lines;", lines[:4])
    json.dump(out, open(os.path.join(HERE, "verbose_golden.json"), "w"), indent=0)
