"""Generate tests/golden/verbose_golden.json from the REAL reference: the stdout of batch_search with QueryParams.verbose = True
(postfilter_vamana.h:155-185,230; the tree classes' own lines around them: range_filter_tree.h:452-457,
super_optimized_postfilter_tree.h:226-267), one thread so that the queries' dumps come in query order.  The figures of the super
tree's two timing lines are replaced by '#'.  Usage: python tests/golden/make_verbose_golden.py"""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
KEEP = ("Starting optimized postfiltering", "Unfiltered return", "Finished a double", "Final frontier size", "Query range", "Testing bucket",
        "Time to find bucket", "Time to do searcht", "Searching bucket")
CASES = {
    # name: (class, constructor keywords, method or None, (beam, mult, max_beam), window fraction exponent[, min_query_to_bucket_ratio])
    "postfilter": ("PostfilterVamanaIndexFloatEuclidian", {}, None, (10, 2, 10000), -5),
    "postfilter_maxbeam": ("PostfilterVamanaIndexFloatEuclidian", {}, None, (8, 3, 40), -7),
    "tree": ("VamanaRangeFilterTreeIndexFloatEuclidian", dict(cutoff=300, split_factor=2), "optimized_postfilter", (10, 2, 10000), -4),
    # the descent's line is printed BEFORE the ratio sends the query to the Fenwick search (range_filter_tree.h:452-466)
    "tree_ratio": ("VamanaRangeFilterTreeIndexFloatEuclidian", dict(cutoff=300, split_factor=2), "optimized_postfilter", (10, 2, 10000), -3, 1.3),
    # the remainders of three_split go through the optimized descent (:517-528)
    "three_split": ("VamanaRangeFilterTreeIndexFloatEuclidian", dict(cutoff=300, split_factor=2), "three_split", (10, 1, 10000), -2),
    # exact-scan leaves: only the descent's lines
    "tree_scan_leaves": ("RangeFilterTreeIndexFloatEuclidian", dict(cutoff=300, split_factor=2), "optimized_postfilter", (10, 1, 10000), -3),
    "super": ("SuperOptimizedPostfilterTreeIndexFloatEuclidian", dict(cutoff=300, split_factor=2, shift_factor=0.5), None, (10, 2, 10000), -4),
    "fenwick": ("VamanaRangeFilterTreeIndexFloatEuclidian", dict(cutoff=300, split_factor=2), "fenwick", (10, 1, 10000), -2),
    # QUIET calls (names ending in _quiet: QueryParams.verbose = False) with windows outside the index's label range: the message of
    # check_empty (range_filter_tree.h:191-203, super_optimized_postfilter_tree.h:173-184) is printed whatever verbose says
    "tree_empty_quiet": ("VamanaRangeFilterTreeIndexFloatEuclidian", dict(cutoff=300, split_factor=2), "optimized_postfilter", (10, 1, 10000), -4),
    "super_empty_quiet": ("SuperOptimizedPostfilterTreeIndexFloatEuclidian", dict(cutoff=300, split_factor=2, shift_factor=0.5), None, (10, 1, 10000), -4),
}
# (the windows of queries 0 and 3 of the *_empty_* cases: above / below every label -- see the worker; tests/test_verbose.py does the same)


def keep(body):
    import re
    out = []
    for l in body.splitlines():
        l = l.strip()
        if l.startswith(KEEP):
            out.append(re.sub(r"\d+ns", "#ns", l) if l.startswith("Time to") else l)
    return out


N, D, NQ, R, L = 2500, 24, 12, 16, 32

WORKER = r'''
import os, sys, json
sys.path.insert(0, os.path.join(%(repo)r, "tests")); sys.path.insert(0, %(repo)r)
import numpy as np
from util import sift_like, distinct_labels, windows
from oracle import oracle as orc
ref = orc.load_reference(prefer=("native", "x86-64-v4"))
cls, kw, method, (beam, mult, maxb), p = %(case)r[:5]
ratio = %(case)r[5] if len(%(case)r) > 5 else None
g = sift_like(%(n)d, %(d)d, 91)
X, Q = g(%(n)d), g(%(nq)d)
labels = distinct_labels(%(n)d, 92)
W = windows(labels, %(nq)d, p, 93)
if "_empty" in %(name)r:
    W[0] = (labels.max() + 1.0, labels.max() + 2.5)
    W[3] = (labels.min() - 5.0, labels.min() - 1.0)
labkw = "filters" if cls.startswith("Postfilter") else "filter_values"
bp = {} if cls.startswith("RangeFilterTree") else dict(build_params=ref.BuildParams(%(r)d, %(l)d, 1.0, ""))
idx = getattr(ref, cls)(X, **{labkw: labels}, **bp, **kw)
sys.stdout.flush()
print("=====BEGIN", flush=True)
a = (Q, W, %(nq)d) + ((method,) if method else ())
idx.batch_search(*a, ref.QueryParams(10, beam, 1.35, 10**7, 10**4, mult, maxb, ratio, not %(name)r.endswith("_quiet")))
'''

if __name__ == "__main__":
    repo = os.path.dirname(os.path.dirname(HERE))
    out = {"inputs": dict(n=N, d=D, nq=NQ, R=R, L=L, seeds=[91, 92, 93]), "cases": {}}
    for name, case in CASES.items():
        code = WORKER % dict(repo=repo, case=case, n=N, d=D, nq=NQ, r=R, l=L, name=name)
        p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PARLAY_NUM_THREADS="1", WANN_NO_TORCH="1"), capture_output=True, text=True)
        assert p.returncode == 0, p.stderr[-2000:]
        body = p.stdout.split("=====BEGIN", 1)[1]
        lines = keep(body)
        assert lines, name
        out["cases"][name] = dict(cls=case[0], kw=case[1], method=case[2], beam=case[3][0], mult=case[3][1], max_beam=case[3][2], fraction=case[4],
                                  ratio=case[5] if len(case) > 5 else None, lines=lines)
        print(name, len(lines), "lines;", lines[:4])
    json.dump(out, open(os.path.join(HERE, "verbose_golden.json"), "w"), indent=0)
