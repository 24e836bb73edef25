"""CPU: the oracle (oracle/restate.cpp) against the golden vectors the REAL reference produced.
This is what pins the oracle (the reference itself ships no vectors for this path)."""
import numpy as np
import pytest

import golden_util as gu


@pytest.mark.parametrize("name", list(gu.FIXTURES))
@pytest.mark.parametrize("kind", list(gu.KINDS))
def test_oracle_replays_reference_outputs(oracle, tmp_path, name, kind):
    n, failures = gu.replay(oracle, name, kind, tmp_path)
    assert n > 0
    assert not failures, "\n".join(failures[:10])


@pytest.mark.parametrize("name", list(gu.FIXTURES))
def test_oracle_quirks(oracle, tmp_path, name):
    idx, data = gu.build_index(oracle, name, "VamanaRangeFilterTreeIndex", tmp_path)
    Q, W = data["Q"], data["W_-3"]
    nq = Q.shape[0]
    # beam >= postfiltering_max_beam: no search at all -> padding (id 0, FLT_MAX)   [SURVEY App. B #4]
    ids, dists = idx.batch_search(Q, W, nq, "optimized_postfilter", oracle.QueryParams(10, 64, 1.35, 10**7, 10**4, 1, 64, None, False))
    assert np.array_equal(ids, data["ids|VamanaRangeFilterTreeIndex|maxbeam"])
    assert np.array_equal(dists, data["dists|VamanaRangeFilterTreeIndex|maxbeam"])
    assert (ids == 0).all() and (dists == np.finfo(np.float32).max).all()
    # doubling overshoots max_beam (8 -> 16 -> 32 > 20): keeps the short result
    ids, dists = idx.batch_search(Q, W, nq, "optimized_postfilter", oracle.QueryParams(10, 8, 1.35, 10**7, 10**4, 1, 20, None, False))
    assert np.array_equal(ids, data["ids|VamanaRangeFilterTreeIndex|overshoot"])
    assert np.array_equal(dists, data["dists|VamanaRangeFilterTreeIndex|overshoot"])


def test_oracle_ratio_fallback_replays_reference_outputs(oracle, tmp_path):
    """min_query_to_bucket_ratio (src/range_filter_tree.h:460-466): ratios 1 / 1.5 / 3 / 8, L2 and inner product"""
    n, failures = gu.replay_ratio(oracle, tmp_path)
    assert n == 48
    assert not failures, "\n".join(failures[:10])
    # the fixture does exercise the branch: a tight ratio changes rows against the no-ratio call of the same batch
    data = next(gu.ratio_cases())[0]
    fx, _ = gu.load("sift_l2")
    assert not np.array_equal(data["ids|sift_l2|1.0|40|2|-3"], fx["ids|VamanaRangeFilterTreeIndex|optimized_postfilter|40|2|-3"]) or \
        not np.array_equal(data["dists|sift_l2|1.0|40|2|-3"], fx["dists|VamanaRangeFilterTreeIndex|optimized_postfilter|40|2|-3"])


def test_hash_and_bits(oracle):
    # parlay::hash64_2 known answers (splitmix64 finaliser) and the seen-filter size rule
    assert oracle.hash64_2(0) == 0
    assert oracle.hash64_2(1) == 0x5692161D100B05E5
    assert [oracle.hash_bits(b) for b in (1, 10, 40, 64, 65, 80, 160, 320, 640, 1280, 10000)] == \
        [10, 10, 10, 10, 11, 11, 13, 15, 17, 19, 25]


# ------------------------------------------------------------------------------------------
# the graph BUILDER against the reference's (tests/golden/build_golden.npz, make_build_golden.py)
# ------------------------------------------------------------------------------------------
def test_insertion_order_is_parlays_random_permutation(oracle):
    import hashlib
    data = gu.load_build()
    seen = 0
    for key in data.files:
        tag, _, n = key.partition("|")
        if tag == "perm":
            assert np.array_equal(oracle.random_permutation(int(n)), data[key]), f"n={n}"
            seen += 1
        elif tag == "perm_sha256":
            got = hashlib.sha256(oracle.random_permutation(int(n)).tobytes()).digest()
            assert got == data[key].tobytes(), f"n={n}"
            seen += 1
    assert seen >= 10


@pytest.mark.parametrize("name", gu.BUILD_CASES)
def test_oracle_builder_writes_the_references_graph_file(oracle, tmp_path, name):
    data = gu.load_build()
    X, (R, L, metric) = data[f"{name}|X"], data[f"{name}|meta"]
    rows = oracle.vamana_build(oracle.pad_rows(X), X.shape[1], int(metric), 0, X.shape[0], int(R), int(L), 1.0, threads=4)
    path = str(tmp_path / "g.bin")
    oracle.graph_save(path, rows)
    assert open(path, "rb").read() == data[f"{name}|file"].tobytes()
