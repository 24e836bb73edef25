"""QueryParams.verbose: the product's dump of every search's doubling loop (stdout, the reference's words:
src/postfilter_vamana.h:155-185,230) and of the tree classes' descent around it (range_filter_tree.h:363-377,452-457,
super_optimized_postfilter_tree.h:226-267; the figures of the super tree's two timing lines masked) against the lines the REAL
reference printed for the same inputs (tests/golden/verbose_golden.json, generator make_verbose_golden.py) -- and the rows of a
verbose call equal a quiet call's.  The *_quiet cases: the message the reference prints for a window outside the index's label
range (range_filter_tree.h:191-203), verbose or not."""
import json
import os
import re
import subprocess
import sys

import pytest

from util import REPO

pytestmark = pytest.mark.gpu
KEEP = ("Starting optimized postfiltering", "Unfiltered return", "Finished a double", "Final frontier size", "Query range", "Testing bucket",
        "Time to find bucket", "Time to do searcht", "Searching bucket")
CHILD = r'''
import os, sys
sys.path.insert(0, os.path.join(%(repo)r, "tests")); sys.path.insert(0, %(repo)r)
import numpy as np
from util import sift_like, distinct_labels, windows
import rangefilteredann_amd, window_ann as wa
c, inp = %(case)r, %(inp)r
g = sift_like(inp["n"], inp["d"], inp["seeds"][0])
X, Q = g(inp["n"]), g(inp["nq"])
labels = distinct_labels(inp["n"], inp["seeds"][1])
W = windows(labels, inp["nq"], c["fraction"], inp["seeds"][2]).astype(np.float32)
if "_empty" in %(name)r:
    W[0] = (labels.max() + 1.0, labels.max() + 2.5)
    W[3] = (labels.min() - 5.0, labels.min() - 1.0)
labkw = "filters" if c["cls"].startswith("Postfilter") else "filter_values"
bp = {} if c["cls"].startswith("RangeFilterTree") else dict(build_params=wa.BuildParams(inp["R"], inp["L"], 1.0, ""))
idx = getattr(wa, c["cls"])(X, **{labkw: labels}, **bp, **c["kw"])
a = (Q, W, inp["nq"]) + ((c["method"],) if c["method"] else ())
loud_flag = not %(name)r.endswith("_quiet")
if loud_flag:
    quiet = idx.batch_search(*a, wa.QueryParams(10, c["beam"], 1.35, 10**7, 10**4, c["mult"], c["max_beam"], c.get("ratio"), False))
sys.stdout.flush()
print("=====BEGIN", flush=True)
loud = idx.batch_search(*a, wa.QueryParams(10, c["beam"], 1.35, 10**7, 10**4, c["mult"], c["max_beam"], c.get("ratio"), loud_flag))
if not loud_flag:
    quiet = loud
sys.stdout.flush()
print("=====END", flush=True)
print("ROWS_EQUAL", bool(np.array_equal(quiet[0], loud[0]) and np.array_equal(quiet[1], loud[1])), flush=True)
'''


@pytest.mark.parametrize("name", ["postfilter", "postfilter_maxbeam", "tree", "tree_ratio", "three_split", "tree_scan_leaves", "super", "fenwick",
                                  "tree_empty_quiet", "super_empty_quiet"])
def test_verbose_dump_equals_the_references(gpu, name):
    gold = json.load(open(os.path.join(REPO, "tests", "golden", "verbose_golden.json")))
    case = gold["cases"][name]
    code = CHILD % dict(repo=REPO, case={k: v for k, v in case.items() if k != "lines"}, inp=gold["inputs"], name=name)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    body = p.stdout.split("=====BEGIN", 1)[1].split("=====END", 1)[0]
    lines = [l.strip() for l in body.splitlines() if l.strip().startswith(KEEP)]
    lines = [re.sub(r"\d+ns", "#ns", l) if l.startswith("Time to") else l for l in lines]
    assert lines == case["lines"], next((i, a, b) for i, (a, b) in enumerate(zip(lines + [""] * 9999, case["lines"] + [""] * 9999)) if a != b)
    assert "ROWS_EQUAL True" in p.stdout
