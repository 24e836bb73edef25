"""Dev tool: from a WANN_TASK_TRACE file (make TRACE=1), what the levels of the speculating tasks found -- for the chains
that had to go beyond their statically speculated levels and for the rest.  Usage: python tools/chain_evidence.py trace.txt k"""
import sys, collections, numpy as np
a = np.loadtxt(sys.argv[1], dtype=np.int64, ndmin=2)
k = int(sys.argv[2]) if len(sys.argv) > 2 else 10
t0 = a[:, 4].min()
subs = collections.defaultdict(dict)   # parent -> {beam: found}
plain = collections.defaultdict(dict)  # task -> {beam: (found, start, end)}
for task, sub, big, beam, st, en, found, parent in a:
    if sub: subs[parent][beam] = found
    else: plain[task][beam] = (found, (st - t0) / 1e5, (en - t0) / 1e5)
beyond, within = [], []
for p, lv in subs.items():
    top = max(lv)
    cont = {b: v for b, v in plain.get(p, {}).items() if b > top}
    (beyond if cont else within).append((p, top, lv, cont))
print(len(subs), "speculating tasks;", len(beyond), "went beyond their highest speculated level")
bytop = collections.defaultdict(list)
for p, top, lv, cont in within: bytop[top].append(lv)
for top in sorted(bytop):
    rows = bytop[top]
    print(f" top {top}: {len(rows)} tasks stayed within; found per level (median / 10th percentile):",
          {b: (int(np.median([r.get(b, -1) for r in rows])), int(np.percentile([r.get(b, -1) for r in rows], 10))) for b in sorted(rows[0])})
for p, top, lv, cont in sorted(beyond, key=lambda x: -max(v[2] for v in x[3].values()))[:12]:
    print(f" task {p}: top {top} found {dict(sorted(lv.items()))} then", {b: (v[0], round(v[1], 2), round(v[2], 2)) for b, v in sorted(cont.items())})

# look-aheads (sub-tasks that started late): what their parent's other levels had found, and when
late = [r for r in a if r[1] and (r[4] - t0) / 1e5 > 1.0 and r[3] >= 2560]
print(len(late), "late big sub-tasks (look-aheads)")
for task, sub, big, beam, st, en, found, parent in sorted(late, key=lambda r: -r[5])[:16]:
    sib = {int(r[3]): (int(r[6]), round((r[4] - t0) / 1e5, 2), round((r[5] - t0) / 1e5, 2)) for r in a if r[1] and r[7] == parent and r[0] != task}
    print(f" look-ahead beam {beam} of task {parent}: start {(st - t0) / 1e5:.2f} end {(en - t0) / 1e5:.2f} found {found}; siblings (found, start, end):", dict(sorted(sib.items())))
