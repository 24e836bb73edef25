import os, sys, time
os.environ.setdefault("WANN_TEST_HOOKS", "1")  # this tool flips WANN_* switches between calls on one index
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np, torch
import window_ann as wa
from util import unit_mixture
n, d, nq = int(sys.argv[1]), 100, 10000
g = unit_mixture(n, d, 2025); X, Q = g(n), g(nq)
labels = ((np.random.default_rng(77).permutation(n) + 0.5) / n).astype(np.float32)
MODE = sys.argv[3] if len(sys.argv) > 3 else ''
def P(*a): print(*a, file=sys.stderr, flush=True)
index = wa.SuperOptimizedPostfilterTreeIndexFloatMips(X, labels, cutoff=1000, split_factor=2, shift_factor=0.5, build_params=wa.BuildParams(64, 500, 1.0, ""))
P("built", sum(index.levels()))
dev = torch.device("cuda:0")
Xt, labt, Qt = torch.from_numpy(X).to(dev), torch.from_numpy(labels).to(dev), torch.from_numpy(Q).to(dev)
ls = np.sort(labels); w = int(n * 2.0 ** -6)
qp = wa.QueryParams(10, 10, 1.35, 10_000_000, 10_000, 1, 10000, None, False)
KEEP = []
S2 = torch.cuda.Stream()
for it in range(int(sys.argv[2])):
    st = np.random.default_rng(it).integers(1, n - w - 1, size=nq)
    Wt = torch.from_numpy(np.stack([ls[st], ls[st + w]], 1).astype(np.float32)).to(dev)
    gt = torch.empty((nq, 10), dtype=torch.int64, device=dev)
    for a in range(0, nq, 256):
        s = -(Qt[a:a + 256] @ Xt.T)
        s.masked_fill_(~((labt[None, :] >= Wt[a:a + 256, 0:1]) & (labt[None, :] <= Wt[a:a + 256, 1:2])), float("inf"))
        gt[a:a + 256] = torch.topk(s, 10, dim=1, largest=False).indices
    del s
    if MODE == 'sync': torch.cuda.synchronize()
    ids_t = torch.empty((nq, 10), dtype=torch.int32, device=dev); dist_t = torch.empty((nq, 10), dtype=torch.float32, device=dev)
    if MODE == 'noengine':
        with torch.cuda.stream(S2):
            ids_t.zero_(); tmp = (Qt @ Qt.T).sum(); S2.synchronize()
    else:
        index.batch_search_device(Qt.data_ptr(), Wt.data_ptr(), nq, 0, "", qp, ids_t.data_ptr(), dist_t.data_ptr(), S2.cuda_stream if MODE == 'tstream' else 0)
    rec = float((gt[:, :, None] == (ids_t.to(torch.int64) & 0xFFFFFFFF)[:, None, :]).any(2).sum(1).double().mean().item() / 10)
    P("iter", it, "recall", round(rec, 4))
    if it < 3:
        for seg in torch.cuda.memory_snapshot(): P('[torch seg] 0x%x + %d' % (seg['address'], seg['total_size']))
        P('ptrs', hex(Qt.data_ptr()), hex(Wt.data_ptr()), hex(ids_t.data_ptr()), hex(Xt.data_ptr()))
    if MODE != 'keep': del ids_t, dist_t, gt, Wt
    else: KEEP.append((ids_t, dist_t, gt, Wt))
