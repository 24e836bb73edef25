"""The experiment driver (rangefilteredann_amd/harness.py) against fixtures produced by the reference's own Python driver
(tests/golden/harness_golden.json, make_harness_golden.py): scoring, early exit, CSV layout, parameter defaults; and an
end-to-end run on the GPU whose recalls must equal the same run driven through the oracle's classes."""
import inspect
import json
import os
import sys

import numpy as np
import pytest

from util import REPO

sys.path.insert(0, REPO)
GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "harness_golden.json")))


@pytest.fixture(scope="module")
def hz():
    from rangefilteredann_amd import harness
    return harness


def test_compute_recall_matches_reference_driver(hz):
    for c in GOLD["compute_recall"]:
        got = hz.compute_recall(np.array(c["results"], dtype=np.uint32), np.array(c["gt"]), c["top_k"])
        assert got == pytest.approx(c["recall"], rel=0, abs=1e-15)


def test_should_break_matches_reference_driver(hz):
    for c in GOLD["should_break"]:
        assert hz.should_break([tuple(r) for r in c["run_results"]]) == c["should_break"], c["run_results"]


def test_results_file_layout(hz, tmp_path):
    g = GOLD["save_results"]
    s = hz.Settings(dataset_folder=str(tmp_path), results_dir=str(tmp_path / "results"), results_file_prefix=g["prefix"], threads=g["threads"])
    ex = hz.Experiments(s)
    tuples = [tuple(t) for t in g["tuples"]]
    for name, text in g["csv"].items():
        path = ex.save_results(tuples, name)
        ex.save_results(tuples[:1], name)
        assert open(path).read() == text
    s.write_results = False
    assert ex.save_results(tuples, "x") is None


def test_defaults_and_constants(hz):
    sig = inspect.signature(hz.build_query_params)
    defaults = {k: v.default for k, v in sig.parameters.items() if v.default is not inspect.Parameter.empty}
    assert defaults == GOLD["build_query_params_defaults"]
    assert list(sig.parameters)[:2] == GOLD["build_query_params_order"][:2]
    c = GOLD["constants"]
    assert (hz.TOP_K, hz.BEAM_SIZES, hz.FINAL_MULTIPLIES, hz.DATASETS) == (c["TOP_K"], c["BEAM_SIZES"], c["FINAL_MULTIPLIES"], c["DATASETS"])
    assert hz.EXPERIMENT_FILTER_WIDTHS == [f"2pow{i}" for i in range(-16, 1)]
    with pytest.raises(Exception, match="Invalid metric"):
        hz.prefilter_index_constructor("cosine", "float")
    with pytest.raises(Exception, match="Invalid data type"):
        hz.vamana_range_filter_tree_constructor("mips", "float16")


def test_cli_without_methods_aborts_like_the_reference(hz, capsys):
    assert hz.main(["--dataset", "sift-128-euclidean"]) == 0
    assert "No experiments specified" in capsys.readouterr().out


@pytest.mark.gpu
@pytest.mark.parametrize("dataset", ["sift-128-euclidean", "glove-100-angular"])
def test_end_to_end_equals_oracle_driven_run(hz, oracle, wa, gpu, tmp_path, monkeypatch, dataset):
    folder = str(tmp_path / "data")
    widths = ["2pow-3", "2pow-6", "2pow0"]
    n, d, nq = 6000, (32 if "sift" in dataset else 20), 64
    hz.write_synthetic_dataset(folder, dataset, n, d, nq, widths, seed=5)
    ranges, gt = hz.get_queries_and_gt(folder, dataset, "2pow-3")
    assert ranges.shape == (nq, 2) and gt.shape == (nq, 10) and (gt >= 0).all()
    methods = ("prefiltering", "postfiltering", "vamana_tree", "optimized_postfiltering", "smart_combined", "three_split", "super_opt_postfiltering")

    def run(tag):
        s = hz.Settings(dataset_folder=folder, cache_root=str(tmp_path / "cache") + "/", results_dir=str(tmp_path / ("results_" + tag)),
                        beam_sizes=[10, 40], final_multiplies=[1, 2], methods=methods, threads=4)
        return hz.Experiments(s).run([dataset], widths)

    got = run("gpu")
    monkeypatch.setattr(hz, "_module", lambda: oracle)  # same driver, oracle classes (they load the cache the engine built)
    want = run("oracle")
    assert list(got) == list(want)
    for key in got:
        # which settings a run visits depends on wall times (should_break's "slower than prefiltering" rule), so compare
        # the settings both runs visited; the first setting of every method is always there
        a, b = {r[1]: r[2] for r in got[key]}, {r[1]: r[2] for r in want[key]}
        common = [name for name in a if name in b]
        assert len(common) >= 7, (key, list(a), list(b))
        # identical recall; integer-valued vectors have exact distance ties whose order at the k-th place is
        # unspecified in the reference (unstable sorts), worth at most a couple of ids per batch
        tol = 0.0 if "angular" in dataset else 3.0 / (nq * 10)
        for name in common:
            assert abs(a[name] - b[name]) <= tol, (key, name, a[name], b[name])
    # exact brute force scores 1.0 against its own ground truth; graph methods reach it at wide windows
    assert all(r[2] == 1.0 for r in got[(dataset, "2pow-3")] if r[1] == "prefiltering")
    assert max(r[2] for r in got[(dataset, "2pow-3")] if r[1].startswith("optimized-postfiltering")) > 0.9
    text = open(os.path.join(str(tmp_path / "results_gpu"), f"{dataset}_results.csv")).read().splitlines()
    assert text[0] + "\n" == hz.RESULTS_HEADER and len(text) == 1 + sum(len(v) for v in got.values())


def test_memory_footprint_tooling(hz, tmp_path, monkeypatch):
    """Counterparts of experiments/all_memories.py / memory_footprint.py: humanize's size strings (decimal units, one
    decimal), the index-type names, the CSV files (header once, rows appended) and the constructor arguments."""
    assert [hz.natural_size(x) for x in (1, 999, 1000, 3_110_000_000, 22_700_000_000)] == ["1 Byte", "999 Bytes", "1.0 kB", "3.1 GB", "22.7 GB"]
    assert hz.MEMORY_INDEX_TYPES == ("postfiltering", "vamana-tree", "super-postfiltering")
    p = hz.write_memory_csv(str(tmp_path / "results"), "memory_usage.csv", ["method", "dataset", "memory"], ["vamana-tree", "sift-128-euclidean", "3.1 GB"])
    hz.write_memory_csv(str(tmp_path / "results"), "memory_usage.csv", ["method", "dataset", "memory"], ["postfiltering", "sift-128-euclidean", "0.8 GB"])
    assert open(p).read().splitlines() == ["method,dataset,memory", "vamana-tree,sift-128-euclidean,3.1 GB", "postfiltering,sift-128-euclidean,0.8 GB"]
    seen = {}

    class Fake:
        def __init__(self, *a, **k):
            seen["a"], seen["k"] = a, k

        def device_bytes(self):
            return 12345

    monkeypatch.setattr(hz, "super_optimized_postfilter_tree_constructor", lambda m, t: Fake)
    monkeypatch.setattr(hz, "postfilter_vamana_constructor", lambda m, t: Fake)
    monkeypatch.setattr(hz, "BuildParams", lambda R, L, alpha, cache: type("BP", (), dict(R=R, L=L, alpha=alpha, cache_path=cache))())
    monkeypatch.chdir(tmp_path)
    _, nb = hz.build_for_memory("super-postfiltering", np.zeros((4, 2), np.float32), np.arange(4, dtype=np.float32), "Euclidian", "ds", 1.0, 2)
    assert nb == 12345 and seen["k"]["cutoff"] == 1000 and seen["k"]["split_factor"] == 2 and seen["k"]["shift_factor"] == 0.5
    bp = seen["k"]["build_params"]
    assert (bp.R, bp.L, bp.alpha) == (64, 500, 1.0) and bp.cache_path.endswith("index_cache/ds-super_opt_postfiltering/")
    hz.build_for_memory("postfiltering", np.zeros((4, 2), np.float32), np.arange(4, dtype=np.float32), "Euclidian", "ds")
    assert len(seen["a"]) == 3 and seen["a"][2].cache_path.endswith("index_cache/ds/unsorted-")
    with pytest.raises(ValueError, match="Invalid index type"):
        hz.build_for_memory("ivf", None, None, "Euclidian", "ds")
