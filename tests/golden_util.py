"""Loading of the golden fixtures (tests/golden/*.npz, produced by tests/golden/make_golden.py from
the real reference) and the loops that replay them against any implementation of the reference's
Python surface (oracle/oracle.py classes or the product's window_ann module)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FIXTURES = {"sift_l2": "FloatEuclidian", "unit_mips": "FloatMips", "u8_l2": "UInt8Euclidian", "i8_mips": "Int8Mips",
            "u8_l2_d512": "UInt8Euclidian",  # (512 bytes per row: beyond what float32 accumulation represents exactly)
            "sift_l2_r96": "FloatEuclidian"}  # (max_degree 96, alpha 1.35: rows of more than 64 neighbours)
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
    R, L, cutoff = [int(x) for x in data["meta"][:3]]
    alpha = float(data["meta"][4]) / 1000.0 if len(data["meta"]) > 4 else 1.0
    cache = unpack_cache(graphs, kind, os.path.join(str(tmpdir), name, kind))
    kw = dict(KINDS[kind])
    if kind.endswith("TreeIndex"):
        kw["cutoff"] = cutoff
    labkw = "filters" if kind == "PostfilterVamanaIndex" else "filter_values"
    cls = getattr(mod, kind + FIXTURES[name])
    idx = cls(data["X"], **{labkw: data["labels"]}, build_params=mod.BuildParams(R, L, alpha, cache), **kw)
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


class RowContext:
    """What a tie-aware comparison needs to VERIFY an id that the expectation does not list: the inputs of the batch.
    metric: "l2" or "mips"; ids are original point numbers (rows of X)."""

    def __init__(self, X, labels, Q, W, metric):
        self.X, self.labels, self.Q, self.W, self.metric = X, np.asarray(labels, dtype=np.float32), Q, np.asarray(W), metric

    def plausible(self, row, pid, dist):
        """pid lies in the row's label window and its (float64) distance to the query is `dist` up to fp32 rounding"""
        if pid < 0 or pid >= len(self.labels):
            return False
        lo, hi = np.float32(self.W[row][0]), np.float32(self.W[row][1])
        if not (lo <= self.labels[pid] <= hi):
            return False
        x, q = self.X[pid].astype(np.float64), self.Q[row].astype(np.float64)
        d64 = -float(x @ q) if self.metric == "mips" else float(((x - q) ** 2).sum())
        return abs(d64 - float(dist)) <= 1e-5 * max(1.0, abs(d64))


def metric_of(class_suffix):
    return "mips" if class_suffix.endswith("Mips") else "l2"


def same_rows(exp_ids, exp_d, got_ids, got_d, tie_aware, ctx=None):
    """Row-wise equality; tie_aware: distances equal exactly and ids equal as multisets inside each run of equal
    distances.  A run cut by the k boundary may be the prefix of a larger tie group of which the expectation shows
    only a part: there every returned id must be one the expectation lists or -- checked from the batch's inputs
    (ctx, required) -- a point inside the query's window at exactly that distance, and no id may repeat more often
    than the expectation repeats ids in that run."""
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
    if ctx is None:
        raise ValueError("tie-aware comparison needs the batch inputs (RowContext) to verify ids at the k boundary")
    k = exp_ids.shape[1]
    for r in np.unique(np.argwhere(exp_ids != got_ids)[:, 0]):
        j = 0
        while j < k:
            e = j
            while e + 1 < k and exp_d[r, e + 1] == exp_d[r, j]:
                e += 1
            exp_run, got_run = sorted(exp_ids[r, j:e + 1].tolist()), sorted(got_ids[r, j:e + 1].tolist())
            if exp_run != got_run:
                if e != k - 1:
                    return False, f"row {r}: ids differ outside a distance tie"
                # run touches the k boundary: unknown members must be verifiable members of the same tie group
                listed = set(exp_run)
                for pid in got_run:
                    if pid not in listed and not ctx.plausible(int(r), int(pid), exp_d[r, j]):
                        return False, f"row {r}: id {pid} at the k boundary is not a window point at distance {exp_d[r, j]!r}"
                max_rep = max(exp_run.count(x) for x in listed)
                if any(got_run.count(x) > max_rep for x in set(got_run)):
                    return False, f"row {r}: an id repeats inside the boundary tie group"
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
        ctx = RowContext(data["X"], data["labels"], Q, W, metric_of(FIXTURES[name]))
        ok, why = same_rows(data["ids|" + key], data["dists|" + key], ids, dists, tie_aware, ctx)
        if not ok:
            failures.append(f"{name} {key}: {why}")
        n += 1
        if max_cases and n >= max_cases:
            break
    return n, failures


def ratio_cases():
    """(fixture, ratio, beam, mult, window key) of tests/golden/ratio_golden.npz (generator: golden/make_ratio_golden.py)."""
    data = np.load(os.path.join(GOLDEN, "ratio_golden.npz"))
    for f in data.files:
        if f.startswith("ids|"):
            name, ratio, beam, mult, p = f[4:].split("|")
            yield data, f[4:], name, float(ratio), int(beam), int(mult), p


def replay_ratio(mod, tmpdir):
    """The min_query_to_bucket_ratio fall-back (src/range_filter_tree.h:460-466) against what the real reference returned."""
    failures, n, built = [], 0, {}
    for data, key, name, ratio, beam, mult, p in ratio_cases():
        if name not in built:
            built[name] = build_index(mod, name, "VamanaRangeFilterTreeIndex", tmpdir)
        idx, fx = built[name]
        Q, W, K = fx["Q"], fx["W_" + p], int(fx["meta"][3])
        qp = mod.QueryParams(K, beam, 1.35, 10_000_000, 10_000, mult, 10000, ratio, False)
        ids, dists = idx.batch_search(Q, W, Q.shape[0], "optimized_postfilter", qp)
        # (queries that fall back to fenwick_tree_search return merged lists: the reference's unstable sort-by-distance)
        ctx = RowContext(fx["X"], fx["labels"], Q, W, metric_of(FIXTURES[name]))
        ok, why = same_rows(data["ids|" + key], data["dists|" + key], ids, dists, True, ctx)
        if not ok:
            failures.append(f"{key}: {why}")
        n += 1
    return n, failures
