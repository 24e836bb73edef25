"""CPU: the plumbing of bench.py's reference legs (tools/ref_legs.py) on a tiny workload -- the rows a leg carries (here: the
oracle's, standing in for the GPU's) must come back as identical to the real reference's, for a graph-search leg and an
exact-scan leg, and the workload table must describe what BASELINE.json names."""
import json
import os
import subprocess
import sys

import numpy as np

from util import REPO


def test_workload_table():
    sys.path.insert(0, REPO)
    import bench
    assert set(bench.WORKLOADS) == {"sift", "deep"}
    s, d = bench.WORKLOADS["sift"], bench.WORKLOADS["deep"]
    assert (s["n"], s["d"], s["split"], s["metric"]) == (1_000_000, 128, 2, "l2")          # BASELINE.json configs[1]
    assert (d["n"], d["d"], d["split"], d["metric"]) == (9_990_000, 96, 4, "mips")         # configs[3]
    assert s["cls"].startswith("VamanaRangeFilterTreeIndex") and d["cls"].endswith("Mips")


def test_reference_legs_worker(oracle, tmp_path):
    sys.path.insert(0, REPO)
    import bench
    from oracle import oracle as orc
    if orc.load_reference() is None:
        import pytest
        pytest.skip("no reference build under oracle/_ref")
    n, d, nq = 3000, 32, 200
    wl = bench.WORKLOADS["sift"]
    X, Q, labels = wl["make"](n, d, nq, 0)
    cache = str(tmp_path / "cache") + "/"
    os.makedirs(cache)
    idx = getattr(orc, wl["cls"])(X, labels, cutoff=wl["cutoff"], split_factor=wl["split"], build_params=orc.BuildParams(wl["R"], wl["L"], wl["alpha"], cache))
    ls = np.sort(labels)
    legs = {}
    for name, p, beam, mult in (("h:20,2", -3, 20, 2), ("f:-9", -9, 10, 1)):
        W = bench.make_windows(ls, nq, p, 7)
        ids, dists = idx.batch_search(Q, W, nq, wl["method"], orc.QueryParams(10, beam, 1.35, 10_000_000, 10_000, mult, 10000, None, False))
        legs["W|" + name], legs["set|" + name] = W, np.array([beam, mult])
        legs["ids|" + name], legs["dists|" + name] = ids, dists
    path = str(tmp_path / "legs.npz")
    np.savez(path, **legs)
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "ref_legs.py"), "--workload", "sift", "--threads", "2", "--n", str(n), "--nq", str(nq),
                        "--dim", str(d), "--cache", cache, "--legs", path, "--seconds", "0.2"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["_kind"] == "reference" and out["_threads"] == 2
    for name in ("h:20,2", "f:-9"):
        assert out[name]["same_dists"] == 1.0 and out[name]["same_id_sets"] == 1.0 and out[name]["qps"] > 0, out[name]
    assert out["h:20,2"]["same_ids"] == 1.0
