"""Generate build_golden.npz from the REAL reference (authoring container only, like make_golden.py).

Pins the graph BUILDER: (1) parlay::random_permutation<int>(n), the insertion order (vamana/index.h:233),
printed by perm_probe.cpp compiled against the parlay headers under /root/reference; (2) the graph cache
files the reference's PostfilterVamanaIndex writes for two inputs with continuous coordinates (no two
candidates of a prune or neighbour sort are exactly equidistant there, so the result does not depend on
libstdc++'s std::sort tie order) and for one integer-valued input full of ties (reproduced by the builders'
reference-tie-order mode: ORC_REF_TIES / WANN_REF_TIES).  Only data is written.  Usage:  python tests/golden/make_build_golden.py
"""
import hashlib
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from util import REPO, quiet_stdout  # noqa: E402

sys.path.insert(0, REPO)
from oracle import oracle as orc  # noqa: E402

PERM_FULL = [1, 2, 3, 5, 100, 1000, 8191, 8192, 8200]
PERM_HASHED = [20000, 65536, 100001, 1000000]
REF_INC = "/root/reference/ParlayANN/parlaylib/include"


def build_case(ref, name, metric, n, d, R, L, seed, out, integer=False):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, d)).astype(np.float32)
    if integer:  # few distinct distances: exact ties in every prune and neighbour sort (libstdc++'s std::sort order decides)
        X = rng.integers(0, 6, size=(n, d)).astype(np.float32)
    if metric == "mips":
        X /= np.linalg.norm(X, axis=1, keepdims=True)
    labels = ((rng.permutation(n) + 0.5) / n).astype(np.float32)
    tmp = tempfile.mkdtemp(prefix="bgold_") + "/"
    cls = ref.PostfilterVamanaIndexFloatMips if metric == "mips" else ref.PostfilterVamanaIndexFloatEuclidian
    with quiet_stdout():
        cls(X, labels, ref.BuildParams(R, L, 1.0, tmp))
    (f,) = os.listdir(tmp)
    out[f"{name}|X"] = X
    out[f"{name}|labels"] = labels
    out[f"{name}|meta"] = np.array([R, L, 1 if metric == "mips" else 0], dtype=np.int64)
    out[f"{name}|file_name"] = np.frombuffer(f.encode(), dtype=np.uint8)
    out[f"{name}|file"] = np.frombuffer(open(tmp + f, "rb").read(), dtype=np.uint8)
    shutil.rmtree(tmp)
    print(name, f, out[f"{name}|file"].size, "bytes")


if __name__ == "__main__":
    ref = orc.load_reference(prefer=("native",))
    assert ref is not None, "build the reference first: make -C oracle ref REF_MARCH=native"
    tmp = tempfile.mkdtemp(prefix="probe_")
    exe = os.path.join(tmp, "perm_probe")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-DHOMEGROWN", "-pthread", "-I" + REF_INC,
                           os.path.join(HERE, "perm_probe.cpp"), "-o", exe])
    out = {}
    for n in PERM_FULL + PERM_HASHED:
        p = np.frombuffer(subprocess.run([exe, str(n)], capture_output=True, check=True).stdout, dtype=np.int32)
        assert p.size == n
        if n in PERM_FULL:
            out[f"perm|{n}"] = p
        else:
            out[f"perm_sha256|{n}"] = np.frombuffer(hashlib.sha256(p.tobytes()).digest(), dtype=np.uint8)
    shutil.rmtree(tmp)
    build_case(ref, "gauss_l2", "l2", 700, 12, 12, 30, 5, out)
    build_case(ref, "unit_mips", "mips", 8200, 6, 6, 14, 6, out)  # n >= 8192: the bucketed permutation
    build_case(ref, "int_l2", "l2", 900, 8, 12, 30, 7, out, integer=True)  # integer-valued vectors: distance ties everywhere
    np.savez_compressed(os.path.join(HERE, "build_golden.npz"), **out)
    print("wrote build_golden.npz:", len(out), "arrays")
