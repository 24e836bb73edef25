"""Experiment driver for the window-search classes: this engine's counterpart of the reference's Python caller
(`experiments/run_our_method.py` + `experiments/wrapper.py`, SURVEY.md 8(a) row a-H).

Same experiment semantics, so result files are comparable line by line with the reference's:

* parameter sweep  beam in BEAM_SIZES x final_beam_multiply in FINAL_MULTIPLIES  (`run_our_method.py:32-33`), `TOP_K` = 10;
* `build_query_params` defaults  (`wrapper.py:334-355`);
* `compute_recall(a, b, top_k)`  = mean_i |set(a_i) & set(b_i[:top_k])| / |set(a_i)|  (`run_our_method.py:174-180`).  The
  reference CALLS it as `compute_recall(results_ids, ground_truth, TOP_K)` (`:265,394,...`), i.e. the denominator is the number of
  DISTINCT returned ids of a query (padding ids collapse to one) -- kept as is so recalls match the reference's files;
* time of a setting = wall time of `batch_search` PLUS `compute_recall` (the result tuple is built in that order, `:391-401`),
  qps = num_queries / time (`:566`);
* `should_break` early exit of the multiplier loop  (`:187-207`);
* result tuples `(filter_width, method_name, recall, time[, build_time, split_factor, memory])` with the reference's method
  names, and the CSV layout of `save_results` (`:540-569`);
* dataset files `<name>.npy`, `<name>_queries.npy`, `<name>_filter-values.npy`, `<name>_queries_<width>_ranges.npy`,
  `<name>_queries_<width>_gt.npy` (`:210-236`); metric = "mips" when the name contains "angular" (`:218`).

Everything runs through the public classes of `window_ann` (the MI355X engine); there is no CPU search path here.
`write_synthetic_dataset` produces a dataset folder in that layout for machines without the real data (ground truth
by exact brute force on the GPU through `PrefilterIndex`).

CLI (same flags as the reference driver):
    python -m rangefilteredann_amd.harness --dataset_folder DIR --dataset sift-128-euclidean --optimized_postfiltering \
        --experiment_filter_width 2pow-3 [--beam_search_size 80] [--num_final_multiplies 1] [--dont_write_to_results_file]
"""
from __future__ import annotations

import argparse
import csv
import gc
import os
import resource
import sys
import time
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

DATASETS = ["sift-128-euclidean", "glove-100-angular", "deep-image-96-angular", "redcaps-512-angular", "adversarial-100-angular"]
EXPERIMENT_FILTER_WIDTHS = [f"2pow{i}" for i in range(-16, 1)]
TOP_K = 10
BEAM_SIZES = [10, 20, 40, 80, 160, 320, 640, 1280]
FINAL_MULTIPLIES = [1, 2, 3, 4, 8, 16, 32]
ALPHAS = [1]
VAMANA_TREE_SPLIT_FACTORS = [2]
SUPER_POSTFILTERING_SPLIT_FACTORS = [2]
SUPER_POSTFILTERING_SHIFT_FACTORS = [0.5]
RESULTS_HEADER = "filter_width,method,recall,average_time,qps,threads\n"  # the reference's header (it has 6 of the 9 columns)


def _module():
    import window_ann  # the engine's drop-in module (rangefilteredann_amd._window_ann)
    return window_ann


# ----------------------------------------------------------------------------------------------------------
# wrapper.py counterparts
# ----------------------------------------------------------------------------------------------------------
_DTYPES = {"float": "Float", "uint8": "Uint8", "int8": "Int8"}
_METRICS = {"Euclidian": "Euclidian", "mips": "Mips"}


def _constructor(prefix: str, metric: str, dtype: str):
    if metric not in _METRICS:
        raise Exception("Invalid metric " + metric)
    if dtype not in _DTYPES:
        raise Exception("Invalid data type " + dtype)
    return getattr(_module(), prefix + _DTYPES[dtype] + _METRICS[metric])


def prefilter_index_constructor(metric, dtype):
    return _constructor("PrefilterIndex", metric, dtype)


def postfilter_vamana_constructor(metric, dtype):
    return _constructor("PostfilterVamanaIndex", metric, dtype)


def vamana_range_filter_tree_constructor(metric, dtype):
    return _constructor("VamanaRangeFilterTreeIndex", metric, dtype)


def super_optimized_postfilter_tree_constructor(metric, dtype):
    return _constructor("SuperOptimizedPostfilterTreeIndex", metric, dtype)


def build_query_params(k, beam_size, cut=1.35, limit=10_000_000, degree_limit=10_000, final_beam_multiply=1,
                       postfiltering_max_beam=10000, min_query_to_bucket_ratio=None, verbose=False):
    return _module().QueryParams(k, beam_size, cut, limit, degree_limit, final_beam_multiply, postfiltering_max_beam,
                                 min_query_to_bucket_ratio, verbose)


def BuildParams(max_degree, limit, alpha, cache_path):
    return _module().BuildParams(max_degree, limit, alpha, cache_path)


# ----------------------------------------------------------------------------------------------------------
# run_our_method.py counterparts: scoring and the early exit
# ----------------------------------------------------------------------------------------------------------
def compute_recall(first, second, top_k):
    """mean over rows of |set(first_i) & set(second_i[:top_k])| / |set(first_i)|  (see the module docstring for how the
    driver calls it)."""
    total = 0.0
    for row_a, row_b in zip(first, second):
        a = set(np.asarray(row_a).tolist())
        b = set(np.asarray(row_b[:top_k]).tolist())
        total += len(a & b) / len(a)
    return total / len(first)


def should_break(run_results) -> bool:
    """Stop increasing final_beam_multiply for the current beam: the last recall exceeds 0.999; or it did not improve on
    the previous entry and the last setting's multiplier (the text after the last '_' of its name) is not 1; or the last
    setting was slower than the most recent prefiltering entry."""
    if not run_results:
        return False
    last = run_results[-1]
    if last[2] > 0.999:
        return True
    if len(run_results) == 1:
        return False
    if last[2] <= run_results[-2][2] and last[1].split("_")[-1] != "1":
        return True
    prefilter_times = [r[3] for r in run_results if r[1] == "prefiltering"]
    return bool(prefilter_times) and last[3] > prefilter_times[-1]


# ----------------------------------------------------------------------------------------------------------
# datasets
# ----------------------------------------------------------------------------------------------------------
def initialize_dataset(folder, dataset_name):
    data = np.load(os.path.join(folder, f"{dataset_name}.npy"))
    queries = np.load(os.path.join(folder, f"{dataset_name}_queries.npy"))
    filter_values = np.load(os.path.join(folder, f"{dataset_name}_filter-values.npy"))
    metric = "mips" if "angular" in dataset_name else "Euclidian"
    return data, queries, filter_values, metric


def get_queries_and_gt(folder, dataset_name, filter_width):
    mid = "_" if filter_width == "" else f"_{filter_width}_"
    ranges = np.load(os.path.join(folder, f"{dataset_name}_queries{mid}ranges.npy"))
    gt = np.load(os.path.join(folder, f"{dataset_name}_queries{mid}gt.npy"))
    return ranges, gt


def write_synthetic_dataset(folder, dataset_name, n, d, num_queries, widths: Sequence[str] = ("2pow-3",), seed=1234,
                            top_k=TOP_K):
    """A dataset folder in the reference's layout from the SURVEY.md 8(d) recipes: 'SIFT-like' integer-valued vectors for
    Euclidean names, unit-norm mixture vectors for '*angular*' names; distinct labels ((perm + 0.5) / n); windows as the
    reference generates them (`generate_datasets/filter_generation_utils.py:11-52`, data-distribution branch): w = int(n * 2^p)
    + 1 consecutive labels with both bounds jittered into the neighbouring gaps, p = 0 -> one window around all labels.
    Ground truth = exact filtered top-k through the engine's brute-force class; -1 where a window holds fewer points."""
    os.makedirs(folder, exist_ok=True)
    rng = np.random.default_rng(seed)
    angular = "angular" in dataset_name
    if angular:
        latent = rng.standard_normal((24, d))
        centres = rng.standard_normal((50, 24))

        def gen(m):
            z = centres[rng.integers(0, 50, m)] + 0.5 * rng.standard_normal((m, 24))
            x = z @ latent + 0.05 * rng.standard_normal((m, d))
            return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)
    else:
        basis = rng.standard_normal((16, d))

        def gen(m):
            z = rng.standard_normal((m, 16))
            return np.clip(np.rint(z @ basis * 18 + 128 + 6 * rng.standard_normal((m, d))), 0, 255).astype(np.float32)
    data, queries = gen(n), gen(num_queries)
    labels = ((rng.permutation(n) + 0.5) / n).astype(np.float32)
    np.save(os.path.join(folder, f"{dataset_name}.npy"), data)
    np.save(os.path.join(folder, f"{dataset_name}_queries.npy"), queries)
    np.save(os.path.join(folder, f"{dataset_name}_filter-values.npy"), labels)
    metric = "mips" if angular else "Euclidian"
    exact = prefilter_index_constructor(metric, "float")(data, labels)
    sorted_labels = np.sort(labels)
    for width in widths:
        p = int(width.replace("2pow", ""))
        if p == 0:  # every point passes (filter_generation_utils.py:18-27)
            ranges = np.tile(np.array([[sorted_labels[0] - rng.integers(1, 100), sorted_labels[-1] + rng.integers(1, 100)]],
                                      dtype=np.float64), (num_queries, 1))
        else:  # w + 1 consecutive labels, bounds jittered into the gaps to the neighbouring labels (:29-52)
            w = int(n * 2.0 ** p)
            start = rng.integers(0, n - w, num_queries)
            end = start + w
            s64 = sorted_labels.astype(np.float64)
            gap_lo = np.where(start > 0, s64[start] - s64[np.maximum(start - 1, 0)], 1.0)
            gap_hi = np.where(end < n - 1, s64[np.minimum(end + 1, n - 1)] - s64[end], 1.0)
            ranges = np.stack([s64[start] - rng.random(num_queries) * gap_lo, s64[end] + rng.random(num_queries) * gap_hi], 1)
        ids, dists = exact.batch_search(queries, ranges, num_queries, build_query_params(top_k, 0))
        ids = ids.astype(np.int64)
        ids[dists == np.finfo(np.float32).max] = -1  # fewer than k points in the window
        mid = "_" if width == "" else f"_{width}_"
        np.save(os.path.join(folder, f"{dataset_name}_queries{mid}ranges.npy"), ranges)
        np.save(os.path.join(folder, f"{dataset_name}_queries{mid}gt.npy"), ids)
    return data, queries, labels


# ----------------------------------------------------------------------------------------------------------
# the experiments
# ----------------------------------------------------------------------------------------------------------
@dataclass
class Settings:
    dataset_folder: str
    cache_root: str = "index_cache"
    results_dir: str = "results"
    results_file_prefix: str = ""
    beam_sizes: List[int] = field(default_factory=lambda: list(BEAM_SIZES))
    final_multiplies: List[int] = field(default_factory=lambda: list(FINAL_MULTIPLIES))
    verbose: bool = False
    write_results: bool = True
    threads: int = os.cpu_count() or 1
    methods: Tuple[str, ...] = ()  # subset of: prefiltering postfiltering vamana_tree optimized_postfiltering smart_combined three_split super_opt_postfiltering


class Experiments:
    """One instance per driver invocation; every `run_*` appends result tuples shaped like the reference's."""

    def __init__(self, settings: Settings):
        self.s = settings
        self._dataset_cache = {}

    # -- helpers
    def _dataset(self, name):
        if name not in self._dataset_cache:
            self._dataset_cache = {name: initialize_dataset(self.s.dataset_folder, name)}
        return self._dataset_cache[name]

    def _cache(self, sub):
        path = os.path.join(self.s.cache_root, sub)
        os.makedirs(path if path.endswith("/") else os.path.dirname(path), exist_ok=True)
        return path

    def _timed(self, all_results, filter_width, name, search, query_gt, extra=()):
        # the reference starts the clock before batch_search and stops it after compute_recall
        start = time.time()
        results = search()
        entry = (filter_width, name, compute_recall(results[0], query_gt, TOP_K), time.time() - start) + tuple(extra)
        all_results.append(entry)
        print(entry, flush=True)

    def _sweep(self, all_results, filter_width, name_of, search_with, query_gt, extra=(), **qp_extra):
        for beam_size in self.s.beam_sizes:
            for mult in self.s.final_multiplies:
                qp = build_query_params(k=TOP_K, beam_size=beam_size, final_beam_multiply=mult, verbose=self.s.verbose, **qp_extra)
                self._timed(all_results, filter_width, name_of(beam_size, mult), lambda: search_with(qp), query_gt, extra)
                if should_break(all_results):
                    break

    # -- experiments (run_our_method.py:239-535)
    def run_prefiltering_experiment(self, all_results, dataset_name, filter_width):
        data, queries, filter_values, metric = self._dataset(dataset_name)
        t0 = time.time()
        index = prefilter_index_constructor(metric, "float")(data, filter_values)
        print(f"Prefiltering index build time: {time.time() - t0:.3f}s", flush=True)
        ranges, gt = get_queries_and_gt(self.s.dataset_folder, dataset_name, filter_width)
        qp = build_query_params(k=TOP_K, beam_size=0, verbose=self.s.verbose)
        self._timed(all_results, filter_width, "prefiltering", lambda: index.batch_search(queries, ranges, queries.shape[0], qp), gt)

    def run_postfiltering_experiment(self, all_results, dataset_name, filter_width, alpha):
        data, queries, filter_values, metric = self._dataset(dataset_name)
        bp = BuildParams(64, 500, alpha, self._cache(f"{dataset_name}/unsorted-"))
        t0 = time.time()
        index = postfilter_vamana_constructor(metric, "float")(data, filter_values, build_params=bp)
        print(f"Naive postfilter build time: {time.time() - t0:.3f}s", flush=True)
        ranges, gt = get_queries_and_gt(self.s.dataset_folder, dataset_name, filter_width)
        self._sweep(all_results, filter_width, lambda b, m: f"postfiltering_{alpha}_{b}_{m}",
                    lambda qp: index.batch_search(queries, ranges, queries.shape[0], qp), gt)

    def run_tree_experiments(self, all_results, dataset_name, filter_width, alpha, split_factor):
        want = [m for m in ("vamana_tree", "optimized_postfiltering", "smart_combined", "three_split") if m in self.s.methods]
        if not want:
            return
        data, queries, filter_values, metric = self._dataset(dataset_name)
        gc.disable()
        rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
        t0 = time.time()
        tree = vamana_range_filter_tree_constructor(metric, "float")(
            data, filter_values, cutoff=1_000, split_factor=split_factor,
            build_params=BuildParams(64, 500, alpha, self._cache(f"{dataset_name}/")))
        build_time = time.time() - t0
        print(f"Vamana tree build time: {build_time:.3f}s", flush=True)
        memory = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss - rss0
        gc.enable()
        ranges, gt = get_queries_and_gt(self.s.dataset_folder, dataset_name, filter_width)
        nq = queries.shape[0]
        extra = (build_time, split_factor, memory)
        if "vamana_tree" in want:
            for beam_size in self.s.beam_sizes:
                qp = build_query_params(k=TOP_K, beam_size=beam_size, verbose=self.s.verbose)
                self._timed(all_results, filter_width, f"vamana-tree_{alpha:.3f}_{split_factor}_{beam_size}",
                            lambda: tree.batch_search(queries, ranges, nq, "fenwick", qp), gt, extra)
        if "optimized_postfiltering" in want:
            self._sweep(all_results, filter_width, lambda b, m: f"optimized-postfiltering_{alpha:.3f}_{split_factor}_{b}_{m}",
                        lambda qp: tree.batch_search(queries, ranges, nq, "optimized_postfilter", qp), gt, extra)
        if "smart_combined" in want:
            self._sweep(all_results, filter_width, lambda b, m: f"smart-combined_{alpha:.3f}_{split_factor}_{b}_{m}",
                        lambda qp: tree.batch_search(queries, ranges, nq, "smart_combined", qp), gt, extra,
                        min_query_to_bucket_ratio=0.05)
        if "three_split" in want:
            self._sweep(all_results, filter_width, lambda b, m: f"three-split_{alpha:.3f}_{split_factor}_{b}_{m}",
                        lambda qp: tree.batch_search(queries, ranges, nq, "three_split", qp), gt, (),
                        min_query_to_bucket_ratio=0.05)

    def run_super_optimized_postfiltering_experiment(self, all_results, dataset_name, filter_width, alpha, split_factor, shift_factor):
        data, queries, filter_values, metric = self._dataset(dataset_name)
        t0 = time.time()
        gc.disable()
        rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
        tree = super_optimized_postfilter_tree_constructor(metric, "float")(
            data, filter_values, cutoff=1_000, split_factor=split_factor, shift_factor=shift_factor,
            build_params=BuildParams(64, 500, alpha, self._cache(f"{dataset_name}-super_opt_postfiltering/")))
        memory = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss - rss0
        gc.enable()
        build_time = time.time() - t0
        print(f"Super optimized postfilter tree build time: {build_time:.3f}s", flush=True)
        ranges, gt = get_queries_and_gt(self.s.dataset_folder, dataset_name, filter_width)
        self._sweep(all_results, filter_width, lambda b, m: f"super-postfiltering_{split_factor}_{shift_factor}_{alpha}_{b}_{m}",
                    lambda qp: tree.batch_search(queries, ranges, queries.shape[0], qp), gt, (build_time, split_factor, memory))

    # -- results file (run_our_method.py:540-569)
    def save_results(self, all_results, dataset_name):
        if not self.s.write_results:
            return None
        os.makedirs(self.s.results_dir, exist_ok=True)
        path = os.path.join(self.s.results_dir, f"{self.s.results_file_prefix}{dataset_name}_results.csv")
        if not os.path.exists(path):
            with open(path, "a") as f:
                f.write(RESULTS_HEADER)
        num_queries = 10000 if "redcaps" not in dataset_name else 800  # the reference's constants, not the batch size
        with open(path, "a") as f:
            for tup in all_results:
                filter_width, name, recall, total_time = tup[:4]
                build_time, branching_factor, memory = (tuple(tup[4:]) + ("", "", ""))[:3]
                f.write(f"{filter_width},{name},{recall},{total_time / num_queries},{num_queries / total_time},{self.s.threads},"
                        f"{build_time},{branching_factor},{memory}\n")
        return path

    # -- the driver loop (run_our_method.py:572-605)
    def run(self, datasets: Sequence[str], filter_widths: Sequence[str], alphas=ALPHAS, split_factors=VAMANA_TREE_SPLIT_FACTORS,
            super_split_factors=SUPER_POSTFILTERING_SPLIT_FACTORS, super_shift_factors=SUPER_POSTFILTERING_SHIFT_FACTORS):
        everything = {}
        for dataset_name in datasets:
            widths = [""] if dataset_name == "adversarial-100-angular" else list(filter_widths)
            for width in widths:
                all_results = []
                if "prefiltering" in self.s.methods:
                    self.run_prefiltering_experiment(all_results, dataset_name, width)
                for alpha in alphas:
                    if "postfiltering" in self.s.methods:
                        self.run_postfiltering_experiment(all_results, dataset_name, width, alpha)
                    for split_factor in split_factors:
                        self.run_tree_experiments(all_results, dataset_name, width, alpha, split_factor)
                    if "super_opt_postfiltering" in self.s.methods:
                        for split_factor in super_split_factors:
                            for shift_factor in super_shift_factors:
                                self.run_super_optimized_postfiltering_experiment(all_results, dataset_name, width, alpha,
                                                                                  split_factor, shift_factor)
                self.save_results(all_results, dataset_name)
                everything[(dataset_name, width)] = all_results
        return everything


# ----------------------------------------------------------------------------------------------------------
# memory footprint (counterparts of experiments/all_memories.py and experiments/memory_footprint.py)
# ----------------------------------------------------------------------------------------------------------
MEMORY_INDEX_TYPES = ("postfiltering", "vamana-tree", "super-postfiltering")  # all_memories.py:108-118


def natural_size(nbytes: int) -> str:
    """humanize.naturalsize (decimal units, one decimal), which all_memories.py:39 applies to the RSS growth."""
    v = float(nbytes)
    if abs(v) < 1000:
        return "%d Bytes" % v if v != 1 else "1 Byte"
    for unit in ("kB", "MB", "GB", "TB", "PB"):
        v /= 1000.0
        if abs(v) < 1000 or unit == "PB":
            return "%.1f %s" % (v, unit)
    return "%.1f PB" % v


def build_for_memory(index_type, data, filter_values, metric, dataset_name, alpha=1.0, split_factor=2, cache_root="index_cache"):
    """Construct the index exactly as all_memories.py:25-83 does and return (index, bytes).  The reference reports the
    growth of the process RSS (its index lives in host memory); this engine's index lives in HBM, so the figure is the
    device footprint (`index.device_bytes()`: vectors + labels + decoding + adjacency pool + partition tables)."""
    if index_type == "postfiltering":
        cons, kw, sub = postfilter_vamana_constructor(metric, "float"), {}, f"{dataset_name}/unsorted-"
    elif index_type == "vamana-tree":
        cons, kw, sub = vamana_range_filter_tree_constructor(metric, "float"), dict(cutoff=1_000, split_factor=split_factor), f"{dataset_name}/"
    elif index_type == "super-postfiltering":
        cons, kw, sub = super_optimized_postfilter_tree_constructor(metric, "float"), dict(cutoff=1_000, split_factor=split_factor, shift_factor=0.5), f"{dataset_name}-super_opt_postfiltering/"
    else:
        raise ValueError("Invalid index type")  # all_memories.py:119-120
    path = os.path.join(cache_root, sub)
    os.makedirs(path if path.endswith("/") else os.path.dirname(path), exist_ok=True)
    bp = BuildParams(64, 500, alpha, path)
    index = cons(data, filter_values, build_params=bp, **kw) if kw else cons(data, filter_values, bp)
    return index, int(index.device_bytes())


def write_memory_csv(results_dir, file_name, headers, row):
    """results/<file_name>, appended, header once (all_memories.py:86-98, memory_footprint.py:41-53)."""
    os.makedirs(results_dir, exist_ok=True)
    path = os.path.join(results_dir, file_name)
    exists = os.path.isfile(path)
    with open(path, "a", newline="") as f:
        w = csv.writer(f)
        if not exists:
            w.writerow(headers)
        w.writerow(row)
    return path


def memory_main(args):
    data, _, filter_values, metric = initialize_dataset(args.dataset_folder, args.dataset)
    if args.index_type:  # all_memories.py: method, dataset, humanised size
        label = {"postfiltering": "postfiltering", "vamana-tree": "vamana-tree", "super-postfiltering": "super postfiltering"}
        _, nbytes = build_for_memory(args.index_type, data, filter_values, metric, args.dataset, 1.0, 2)
        print(write_memory_csv("results", "memory_usage.csv", ["method", "dataset", "memory"], [label[args.index_type], args.dataset, natural_size(nbytes)]))
    else:  # memory_footprint.py: method, branching factor, KiB (ru_maxrss units)
        b = args.vamana_tree_split_factor
        _, nbytes = build_for_memory("vamana-tree", data, filter_values, metric, args.dataset, args.alpha if args.alpha is not None else 1.0, b)
        print(write_memory_csv("results", "vamana_tree_memory_usage.csv", ["method", "branching_factor", "memory"], ["vamana-tree", b, nbytes // 1024]))
    return 0


ALL_METHODS = ("prefiltering", "postfiltering", "vamana_tree", "optimized_postfiltering", "smart_combined", "three_split",
               "super_opt_postfiltering")


def main(argv=None):
    ap = argparse.ArgumentParser(description="window-search experiments on the MI355X engine (flags of experiments/run_our_method.py)")
    ap.add_argument("--dataset_folder", default=os.environ.get("WANN_DATASET_FOLDER", "datasets"))
    ap.add_argument("--threads", type=int, default=None, help="host threads for index construction (PARLAY_NUM_THREADS)")
    for m in ALL_METHODS:
        ap.add_argument(f"--{m}", action="store_true")
    ap.add_argument("--all_methods", action="store_true")
    ap.add_argument("--results_file_prefix", default="")
    ap.add_argument("--beam_search_size", type=int, default=None)
    ap.add_argument("--experiment_filter_width", type=str, default=None)
    ap.add_argument("--num_final_multiplies", type=int, default=None)
    ap.add_argument("--dataset", type=str, default=None)
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--dont_write_to_results_file", action="store_true")
    ap.add_argument("--vamana_tree_split_factor", type=int)
    ap.add_argument("--alpha", type=float)
    ap.add_argument("--super_opt_postfiltering_split_factor", type=float)
    ap.add_argument("--super_opt_postfiltering_shift_factor", type=float)
    ap.add_argument("--synthetic", type=str, default=None, metavar="N,D,NQ",
                    help="write a synthetic dataset of that shape into --dataset_folder first (no real data on this machine)")
    ap.add_argument("--memory", action="store_true",
                    help="index memory footprint instead of a search experiment: with --index_type {postfiltering,vamana-tree,"
                         "super-postfiltering} as experiments/all_memories.py, else with --vamana_tree_split_factor as "
                         "experiments/memory_footprint.py")
    ap.add_argument("--index_type", type=str, default=None)
    args = ap.parse_args(argv)
    threads = args.threads or (os.cpu_count() or 1)
    os.environ["PARLAY_NUM_THREADS"] = str(threads)
    if args.memory:
        if not args.dataset or (not args.index_type and args.vamana_tree_split_factor is None):
            ap.error("--memory needs --dataset and --index_type or --vamana_tree_split_factor")
        if args.synthetic:
            n, d, nq = (int(x) for x in args.synthetic.split(","))
            write_synthetic_dataset(args.dataset_folder, args.dataset, n, d, nq, ["2pow-3"])
        return memory_main(args)
    methods = ALL_METHODS if args.all_methods else tuple(m for m in ALL_METHODS if getattr(args, m))
    if not methods:
        print("NOTE: No experiments specified, so aborting")
        ap.print_help()
        return 0
    datasets = [args.dataset] if args.dataset else list(DATASETS)
    widths = [args.experiment_filter_width] if args.experiment_filter_width else list(EXPERIMENT_FILTER_WIDTHS)
    if args.synthetic:
        n, d, nq = (int(x) for x in args.synthetic.split(","))
        for name in datasets:
            write_synthetic_dataset(args.dataset_folder, name, n, d, nq, [""] if name == "adversarial-100-angular" else widths)
    settings = Settings(dataset_folder=args.dataset_folder, results_file_prefix=args.results_file_prefix,
                        beam_sizes=[args.beam_search_size] if args.beam_search_size else list(BEAM_SIZES),
                        final_multiplies=[args.num_final_multiplies] if args.num_final_multiplies else list(FINAL_MULTIPLIES),
                        verbose=args.verbose, write_results=not args.dont_write_to_results_file, threads=threads, methods=methods)
    Experiments(settings).run(
        datasets, widths,
        alphas=[args.alpha] if args.alpha is not None else ALPHAS,
        split_factors=[args.vamana_tree_split_factor] if args.vamana_tree_split_factor is not None else VAMANA_TREE_SPLIT_FACTORS,
        super_split_factors=[args.super_opt_postfiltering_split_factor] if args.super_opt_postfiltering_split_factor is not None else SUPER_POSTFILTERING_SPLIT_FACTORS,
        super_shift_factors=[args.super_opt_postfiltering_shift_factor] if args.super_opt_postfiltering_shift_factor is not None else SUPER_POSTFILTERING_SHIFT_FACTORS)
    return 0


if __name__ == "__main__":
    sys.exit(main())
