"""Golden vectors for the unfiltered VamanaIndex API (python_bindings.cpp:92-109, ParlayANN/python/vamana_index.cpp,
builder.cpp), produced by the REAL reference module built from /root/reference (oracle/_ref): point files in, graph
files written by build_vamana_*_index, batch_search outputs for several (knn, beam_width) incl. the k / cut step of
beamSearch.h:159-167.  Run in the authoring container with free() disabled (the reference index reads a buffer it has already freed, see nofree.c):
    gcc -shared -fPIC tests/golden/nofree.c -o /tmp/nofree.so && LD_PRELOAD=/tmp/nofree.so python tests/golden/make_vamana_golden.py"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
os.environ["WANN_NO_TORCH"] = "1"
from oracle import oracle as orc  # noqa: E402
from util import quiet_stdout, sift_like, unit_mixture  # noqa: E402

ref = orc.load_reference()
assert ref is not None, "build the reference first: make -C oracle ref"


def write_bin(path, X):
    with open(path, "wb") as f:
        np.array(X.shape, dtype=np.uint32).tofile(f)
        X.tofile(f)


out = {}
keep = []  # (never destroyed: see the end of the script)
tmp = tempfile.mkdtemp(prefix="vamana_golden_")
CASES = [("l2", "float_euclidian", "VamanaFloatEuclidianIndex", sift_like(2500, 24, 5), np.float32, 24),
         ("mips", "float_mips", "VamanaFloatMipsIndex", unit_mixture(2500, 20, 6), np.float32, 20),
         ("u8", "uint8_euclidian", "VamanaUInt8EuclidianIndex", sift_like(1500, 32, 7), np.uint8, 32)]
for name, lower, cls, gen, dt, d in CASES:
    n, nq, R, L, alpha = (2500 if dt == np.float32 else 1500), 40, 16, 32, 1.2
    X, Q = gen(n).astype(dt), gen(nq).astype(dt)
    data, graph = os.path.join(tmp, name + ".bin"), os.path.join(tmp, name + ".graph")
    write_bin(data, X)
    with quiet_stdout():
        getattr(ref, "build_vamana_" + lower + "_index")("ignored", data, graph, R, L, alpha)
        idx = getattr(ref, cls)(data, graph, n, d)  # (first argument = point file: see the binding's argument names)
    keep.append(idx)
    out[name + "/X"], out[name + "/Q"] = X, Q
    out[name + "/graph"] = np.fromfile(graph, dtype=np.uint8)
    out[name + "/meta"] = np.array([R, L, int(alpha * 1000)], dtype=np.int64)
    for knn, beam in ((1, 8), (10, 10), (10, 40), (5, 100), (10, 200)):
        with quiet_stdout():
            ids, dists = idx.batch_search(Q, nq, knn, beam)
        out[f"{name}/ids|{knn}|{beam}"], out[f"{name}/dists|{knn}|{beam}"] = ids, dists
np.savez_compressed(os.path.join(HERE, "vamana_golden.npz"), **out)
print("wrote", os.path.join(HERE, "vamana_golden.npz"), os.path.getsize(os.path.join(HERE, "vamana_golden.npz")), "bytes", flush=True)
# The reference's VamanaIndex copy-assigns a PointRange that owns a raw buffer (vamana_index.cpp:47-48, point_range.h:113-115):
# its destructor frees that buffer a second time.  The outputs above are complete; leave without running destructors.
os._exit(0)
