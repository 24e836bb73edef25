"""Generate harness_golden.json from the reference's own Python driver (authoring container only).

`experiments/run_our_method.py` parses sys.argv and exits at import, so its functions are taken out of the file with
`ast` and executed here, in this process, on seeded random inputs; only the inputs and the outputs are stored:
`compute_recall`, `should_break`, the CSV text `save_results` writes, and the defaults of `wrapper.build_query_params`.
Usage:  python tests/golden/make_harness_golden.py
"""
import ast
import inspect
import json
import os
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/experiments"


def functions_of(path, names, namespace):
    tree = ast.parse(open(path).read())
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert len(keep) == len(names), [n.name for n in keep]
    exec(compile(ast.Module(body=keep, type_ignores=[]), path, "exec"), namespace)
    return namespace


def method_name(rng):
    kind = rng.integers(0, 5)
    beam, mult = int(rng.choice([10, 20, 40, 80])), int(rng.choice([1, 1, 2, 3, 4, 8, 16, 32]))
    return ["prefiltering", f"postfiltering_1_{beam}_{mult}", f"optimized-postfiltering_1.000_2_{beam}_{mult}",
            f"three-split_1.000_2_{beam}_{mult}", f"super-postfiltering_2_0.5_1_{beam}_{mult}"][kind]


if __name__ == "__main__":
    rng = np.random.default_rng(2024)
    out = {}
    ns = functions_of(os.path.join(REF, "run_our_method.py"), ["compute_recall", "should_break"], {})
    cases = []
    for c in range(12):
        nq, k = int(rng.integers(1, 40)), 10
        res = rng.integers(0, 60, size=(nq, k)).astype(np.uint32)
        res[rng.random(nq) < 0.3, rng.integers(3, k):] = 0  # padded rows (id 0 repeated)
        gt = rng.integers(0, 60, size=(nq, int(rng.choice([10, 12, 100])))).astype(np.int64)
        top_k = int(rng.choice([10, 5]))
        cases.append(dict(results=res.tolist(), gt=gt.tolist(), top_k=top_k, recall=ns["compute_recall"](res, gt, top_k)))
    out["compute_recall"] = cases
    cases = []
    for c in range(400):
        m = int(rng.integers(0, 6))
        runs = []
        for _ in range(m):
            recall = float(rng.choice([0.5, 0.9, 0.95, 0.9991, 1.0, 0.999, 0.97]))
            runs.append(["2pow-3", method_name(rng), recall, float(rng.choice([0.5, 1.0, 2.0, 4.0]))])
        cases.append(dict(run_results=runs, should_break=bool(ns["should_break"]([tuple(r) for r in runs]))))
    out["should_break"] = cases
    # save_results: the CSV text for result tuples of every arity the driver produces
    tmp = tempfile.mkdtemp()
    cwd = os.getcwd()
    os.chdir(tmp)
    os.makedirs("results")
    ns2 = dict(os=os, num_threads=48, args=types.SimpleNamespace(results_file_prefix="pre_", dont_write_to_results_file=False))
    functions_of(os.path.join(REF, "run_our_method.py"), ["save_results"], ns2)
    tuples = [("2pow-3", "prefiltering", 1.0, 2.5),
              ("2pow-3", "optimized-postfiltering_1.000_2_10_1", 0.9375, 0.125, 17.5, 2, 123456),
              ("2pow-3", "three-split_1.000_2_20_2", 0.5, 0.75),
              ("", "super-postfiltering_2_0.5_1_40_1", 0.99, 3.0, 100.25, 2, 42)]
    csv = {}
    for name in ("sift-128-euclidean", "redcaps-512-angular"):
        ns2["save_results"](tuples, name)
        ns2["save_results"](tuples[:1], name)  # second call appends without a second header
        csv[name] = open(f"results/pre_{name}_results.csv").read()
    os.chdir(cwd)
    out["save_results"] = dict(tuples=[list(t) for t in tuples], threads=48, prefix="pre_", csv=csv)
    # wrapper.build_query_params defaults
    ns3 = dict(QueryParams=lambda *a: a)
    functions_of(os.path.join(REF, "wrapper.py"), ["build_query_params"], ns3)
    sig = inspect.signature(ns3["build_query_params"])
    out["build_query_params_defaults"] = {k: v.default for k, v in sig.parameters.items() if v.default is not inspect.Parameter.empty}
    out["build_query_params_order"] = list(ns3["build_query_params"](k="k", beam_size="beam_size"))
    # the driver's sweep constants
    tree = ast.parse(open(os.path.join(REF, "run_our_method.py")).read())
    consts = {}
    for node in tree.body:
        if isinstance(node, ast.Assign) and isinstance(node.targets[0], ast.Name) and node.targets[0].id in ("TOP_K", "BEAM_SIZES", "FINAL_MULTIPLIES", "DATASETS"):
            consts[node.targets[0].id] = ast.literal_eval(node.value)
    out["constants"] = consts
    json.dump(out, open(os.path.join(HERE, "harness_golden.json"), "w"))
    print("wrote harness_golden.json:", {k: (len(v) if hasattr(v, "__len__") else v) for k, v in out.items()})
