"""Dev tool: raw beam search of the product against the oracle over (env, beam, graph flavour); prints mismatch summaries."""
import os, sys, numpy as np
os.environ.setdefault("WANN_TEST_HOOKS", "1")  # this tool flips WANN_* switches between calls on one index
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import rangefilteredann_amd, window_ann as wa
from oracle import oracle
from util import sift_like, unit_mixture
oracle.build()

def h2(x):
    m = (1 << 64) - 1; x &= m
    x = ((x ^ (x >> 30)) * 0xbf58476d1ce4e5b9) & m; x = ((x ^ (x >> 27)) * 0x94d049bb133111eb) & m
    return x ^ (x >> 31)

n, nq, R, L = 6000, 64, 32, 64
start, sn = 300, 5000
for metric, gen, d in ((0, sift_like, 128), (1, unit_mixture, 100)):
    g = gen(n, d, 12); X, Q = g(n), g(nq); Xp = oracle.pad_rows(X)
    rows0 = oracle.vamana_build(Xp, d, metric, start, sn, R, L, 1.0).copy()
    rows1 = rows0.copy()
    same = [b for b in range(1, sn) if (h2(b) ^ h2(0)) & 1023 == 0]
    rng = np.random.default_rng(3)
    for r in rng.choice(sn, 300, replace=False):
        b = same[int(r) % len(same)]; rows1[r, 0] = max(int(rows1[r, 0]), 3); rows1[r, 1:4] = (0, b, 0)
    qids = np.arange(nq, dtype=np.int64) + 10**6
    for gname, rows in (("plain", rows0), ("twice", rows1)):
        exp = {}
        for beam in (16, 40, 100, 160, 300, 1000):
            exp[beam] = [oracle.beam_search(rows, Xp, d, metric, start, Q[i], int(qids[i]), beam) for i in range(nq)]
        for env in ({}, {"WANN_FORCE_GENERAL": "1"}, {"WANN_OLD_GENERAL": "1"}, {"WANN_RAW_BIG_LDS": "1"}, {"WANN_RAW_BIG_LDS": "1", "WANN_FORCE_GENERAL": "1"}):
            for k in ("WANN_FORCE_GENERAL", "WANN_OLD_GENERAL", "WANN_RAW_BIG_LDS"): os.environ.pop(k, None)
            os.environ.update(env)
            for beam in (16, 40, 100, 160, 300, 1000):
                ids, dists, sizes, hops, cmps = wa.raw_beam_search(metric, X, rows, start, Q, qids, beam)
                bad_ids = bad_h = bad_c = 0; first = None
                for i in range(nq):
                    oi, od, vi, vd, dc = exp[beam][i]
                    m = int(sizes[i])
                    ok = m == len(oi) and np.array_equal(ids[i, :m], oi) and np.array_equal(dists[i, :m], od)
                    if not ok:
                        bad_ids += 1
                        if first is None:
                            k0 = next((j for j in range(min(m, len(oi))) if ids[i, j] != oi[j] or dists[i, j] != od[j]), min(m, len(oi)))
                            first = (i, m, len(oi), k0, ids[i, max(0,k0-1):k0+2].tolist(), oi[max(0,k0-1):k0+2].tolist(), dists[i, max(0,k0-1):k0+2].tolist(), od[max(0,k0-1):k0+2].tolist())
                    bad_h += int(hops[i]) != len(vi); bad_c += int(cmps[i]) != dc
                if bad_ids or bad_h or bad_c:
                    print(f"metric {metric} graph {gname} env {env} beam {beam}: bad rows {bad_ids} hops {bad_h} cmps {bad_c} first {first}", flush=True)
    print("metric", metric, "done", flush=True)
