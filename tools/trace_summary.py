"""Dev tool: summarise a WANN_TASK_TRACE file (task sub big beam start end found parent; 100 MHz ticks)."""
import sys, numpy as np
a = np.loadtxt(sys.argv[1], dtype=np.int64, ndmin=2)
t0 = a[:, 4].min(); st = (a[:, 4] - t0) / 1e5; en = (a[:, 5] - t0) / 1e5   # ms
print(f"{len(a)} searches, span {en.max():.2f} ms")
for b in sorted(set(a[:, 3])):
    m = a[:, 3] == b
    print(f"  beam {b:5d}: {m.sum():6d} searches (sub {a[m,1].sum():5d} big {a[m,2].sum():4d})  dur ms mean {np.mean(en[m]-st[m]):7.3f} max {np.max(en[m]-st[m]):7.3f}  start max {st[m].max():6.2f}  end max {en[m].max():6.2f}")
o = np.argsort(-en)[:12]
print("latest finishers:")
for i in o:
    print(f"   task {a[i,0]} sub {a[i,1]} big {a[i,2]} beam {a[i,3]} start {st[i]:.2f} end {en[i]:.2f}")
# occupancy over time: searches in flight per 0.1 ms
import collections
bins = collections.Counter()
for s_, e_ in zip(st, en):
    for b in range(int(s_ * 10), int(e_ * 10) + 1):
        bins[b] += 1
print("in-flight searches per 0.1 ms:", " ".join(str(bins[b]) for b in range(0, int(en.max() * 10) + 1)))
