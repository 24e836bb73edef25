"""QueryParams.verbose: the product's dump of every search's doubling loop (stdout, the reference's words:
src/postfilter_vamana.h:155-185,230) against the lines the REAL reference printed for the same inputs
(tests/golden/verbose_golden.json, generator make_verbose_golden.py) -- and the rows of a verbose call equal a quiet call's."""
import json
import os
import subprocess
import sys

import pytest

from util import REPO

pytestmark = pytest.mark.gpu
KEEP = ("Starting optimized postfiltering", "Unfiltered return", "Finished a double", "Final frontier size")
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
labkw = "filters" if c["cls"].startswith("Postfilter") else "filter_values"
idx = getattr(wa, c["cls"])(X, **{labkw: labels}, build_params=wa.BuildParams(inp["R"], inp["L"], 1.0, ""), **c["kw"])
a = (Q, W, inp["nq"]) + ((c["method"],) if c["method"] else ())
quiet = idx.batch_search(*a, wa.QueryParams(10, c["beam"], 1.35, 10**7, 10**4, c["mult"], c["max_beam"], None, False))
sys.stdout.flush()
print("=====BEGIN", flush=True)
loud = idx.batch_search(*a, wa.QueryParams(10, c["beam"], 1.35, 10**7, 10**4, c["mult"], c["max_beam"], None, True))
sys.stdout.flush()
print("=====END", flush=True)
print("ROWS_EQUAL", bool(np.array_equal(quiet[0], loud[0]) and np.array_equal(quiet[1], loud[1])), flush=True)
'''


@pytest.mark.parametrize("name", ["postfilter", "postfilter_maxbeam", "tree"])
def test_verbose_dump_equals_the_references(gpu, name):
    gold = json.load(open(os.path.join(REPO, "tests", "golden", "verbose_golden.json")))
    case = gold["cases"][name]
    code = CHILD % dict(repo=REPO, case={k: v for k, v in case.items() if k != "lines"}, inp=gold["inputs"])
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    body = p.stdout.split("=====BEGIN", 1)[1].split("=====END", 1)[0]
    lines = [l.strip() for l in body.splitlines() if l.strip().startswith(KEEP)]
    assert lines == case["lines"], next((i, a, b) for i, (a, b) in enumerate(zip(lines + [""] * 9999, case["lines"] + [""] * 9999)) if a != b)
    assert "ROWS_EQUAL True" in p.stdout
