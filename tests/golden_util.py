"""Loading of the golden fixtures (tests/golden/*.npz, produced by tests/golden/make_golden.py from
the real reference) and the loops that replay them against any implementation of the reference's
Python surface (oracle/oracle.py classes or the product's window_ann module)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FIXTURES = {"sift_l2": "FloatEuclidian", "unit_mips": "FloatMips", "u8_l2": "UInt8Euclidian", "i8_mips": "Int8Mips"}
KINDS = {
    "VamanaRangeFilterTreeIndex": dict(split_factor=2),
    "SuperOptimizedPostfilterTreeIndex": dict(split_factor=2, shift_factor=0.5),
    "PostfilterVamanaIndex": dict(),
    "RangeFilterTreeIndex": dict(split_factor=2),
    "PrefilterIndex": dict(),
}
TIE_AWARE_KINDS = ("PrefilterIndex", "RangeFilterTreeIndex")  # brute-force results: unstable sort in the reference


def load(name):
    data = np.load(os.path.join(GOLDEN, name + ".npz"))
    graphs = np.load(os.path.join(GOLDEN, name + "_graphs.npz"))
    return data, graphs


BUILD_CASES = ["gauss_l2", "unit_mips"]


def load_build():
    """Builder fixtures: parlay's insertion permutations and two graph files written by the reference."""
    return np.load(os.path.join(GOLDEN, "build_golden.npz"))


def unpack_cache(graphs, kind, dst):
    """Write the reference-built graph cache files of `kind` under dst/ and return the prefix."""
    os.makedirs(dst, exist_ok=True)
    for key in graphs.files:
        k, fn = key.split("/", 1)
        if k == kind:
            with open(os.path.join(dst, fn), "wb") as f:
                f.write(graphs[key].tobytes())
    return dst.rstrip("/") + "/"


def build_index(mod, name, kind, tmpdir):
    data, graphs = load(name)
    R, L, cutoff, _ = [int(x) for x in data["meta"]]
    cache = unpack_cache(graphs, kind, os.path.join(str(tmpdir), name, kind))
    kw = dict(KINDS[kind])
    if kind.endswith("TreeIndex"):
        kw["cutoff"] = cutoff
    labkw = "filters" if kind == "PostfilterVamanaIndex" else "filter_values"
    cls = getattr(mod, kind + FIXTURES[name])
    idx = cls(data["X"], **{labkw: data["labels"]}, build_params=mod.BuildParams(R, L, 1.0, cache), **kw)
    return idx, data


def cases(data, kind):
    """Yield (key, method, beam, mult, window_key) for every stored expectation of `kind`."""
    for f in data.files:
        if not f.startswith("ids|" + kind + "|"):
            continue
        parts = f.split("|")
        if len(parts) != 6:
            continue
        _, _, method, beam, mult, p = parts
        yield f[4:], method, int(beam), int(mult), p


def same_rows(exp_ids, exp_d, got_ids, got_d, tie_aware):
    """Row-wise equality; tie_aware: distances equal exactly and ids equal as multisets inside each
    run of equal distances, except a run cut by the k boundary where only membership in the
    candidate set can be checked."""
    if exp_ids.shape != got_ids.shape:
        return False, "shape"
    if not np.array_equal(exp_d, got_d):
        bad = np.argwhere(exp_d != got_d)[0]
        return False, f"dist mismatch at {tuple(bad)}: {exp_d[tuple(bad)]!r} vs {got_d[tuple(bad)]!r}"
    if np.array_equal(exp_ids, got_ids):
        return True, ""
    if not tie_aware:
        bad = np.argwhere(exp_ids != got_ids)[0]
        return False, f"id mismatch at {tuple(bad)}: {exp_ids[bad[0]]} vs {got_ids[bad[0]]}"
    k = exp_ids.shape[1]
    for r in np.unique(np.argwhere(exp_ids != got_ids)[:, 0]):
        j = 0
        while j < k:
            e = j
            while e + 1 < k and exp_d[r, e + 1] == exp_d[r, j]:
                e += 1
            if e == k - 1:  # run touches the k boundary: it may be the prefix of a larger tie group
                pass
            elif sorted(exp_ids[r, j:e + 1]) != sorted(got_ids[r, j:e + 1]):
                return False, f"row {r}: ids differ outside a distance tie"
            j = e + 1
    return True, ""


def replay(mod, name, kind, tmpdir, max_cases=None):
    idx, data = build_index(mod, name, kind, tmpdir)
    K = int(data["meta"][3])
    Q = data["Q"]
    nq = Q.shape[0]
    n = 0
    failures = []
    for key, method, beam, mult, p in cases(data, kind):
        W = data["W_" + p]
        qp = mod.QueryParams(K, beam, 1.35, 10_000_000, 10_000, mult, 10000, None, False)
        args = (Q, W, nq) + ((method,) if kind.endswith("RangeFilterTreeIndex") else ())
        ids, dists = idx.batch_search(*args, qp)
        # merged / brute-forced results go through the reference's unstable sort-by-distance
        tie_aware = kind in TIE_AWARE_KINDS or method in ("fenwick", "three_split") or p in ("-7", "edge")
        ok, why = same_rows(data["ids|" + key], data["dists|" + key], ids, dists, tie_aware)
        if not ok:
            failures.append(f"{name} {key}: {why}")
        n += 1
        if max_cases and n >= max_cases:
            break
    return n, failures
