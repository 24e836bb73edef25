"""Generate tests/golden/ratio_golden.npz from the REAL reference: the `min_query_to_bucket_ratio` fall-back of
optimized_postfiltering_search (src/range_filter_tree.h:460-466 -- a window that is a small share of its smallest containing
bucket goes to fenwick_tree_search instead).

Runs only in the authoring container (reference module from `make -C oracle ref REF_MARCH=native`).  The reference LOADS the
graph files of the existing fixtures (sift_l2 / unit_mips, *_graphs.npz), so only DATA is written: the (ids, dists) it returned
per (fixture, ratio, beam, multiplier, window fraction).  Usage: python tests/golden/make_ratio_golden.py
"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import golden_util as gu  # noqa: E402
from util import REPO, quiet_stdout  # noqa: E402

sys.path.insert(0, REPO)
from oracle import oracle as orc  # noqa: E402

RATIOS = [1.0, 1.5, 3.0, 8.0]
SETTINGS = [(10, 1), (40, 2)]
FRACTIONS = ["-5", "-3", "-1"]

if __name__ == "__main__":
    ref = orc.load_reference(prefer=("native",))
    assert ref is not None, "build the reference first: make -C oracle ref REF_MARCH=native"
    out = {}
    with tempfile.TemporaryDirectory(prefix="ratio_golden_") as tmp:
        for name in ("sift_l2", "unit_mips"):
            with quiet_stdout():
                idx, data = gu.build_index(ref, name, "VamanaRangeFilterTreeIndex", tmp)
            Q, K = data["Q"], int(data["meta"][3])
            for ratio in RATIOS:
                for beam, mult in SETTINGS:
                    for p in FRACTIONS:
                        qp = ref.QueryParams(K, beam, 1.35, 10_000_000, 10_000, mult, 10000, ratio, False)
                        with quiet_stdout():
                            ids, dists = idx.batch_search(Q, data["W_" + p], Q.shape[0], "optimized_postfilter", qp)
                        key = f"{name}|{ratio}|{beam}|{mult}|{p}"
                        out["ids|" + key], out["dists|" + key] = ids, dists
    np.savez_compressed(os.path.join(HERE, "ratio_golden.npz"), **out)
    print("ratio_golden.npz:", len(out) // 2, "cases")
