import os, sys, time, numpy as np, shutil
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from util import sift_like
import window_ann as wa
n = int(sys.argv[1])
X = sift_like(n, 128, 1234)(n)
rng = np.random.default_rng(4321)
labels = ((rng.permutation(n) + 0.5) / n).astype(np.float32)
t = time.time()
idx = wa.VamanaRangeFilterTreeIndexFloatEuclidian(X, labels, cutoff=1000, split_factor=2, build_params=wa.BuildParams(64, 500, 1.0, ""))
print("n", n, "gpu build+upload", round(time.time() - t, 1), "s", idx.levels(), flush=True)
