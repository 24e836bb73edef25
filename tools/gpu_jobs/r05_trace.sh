#!/bin/bash
# round 5 (late): timelines of 2^-5 ... 2^-7 (which chain ends the batch?) from a make TRACE=1 build
export TMPDIR=/tmp
O=gpurun_out/r05trace
mkdir -p $O
export LD_LIBRARY_PATH=$PWD/tools/_scratch/trace:$LD_LIBRARY_PATH
for p in ${FRACS:--6 -7 -5}; do
  WANN_TASK_TRACE=$O/trace$p.txt python tools/frac_probe.py --fractions=$p --settings 80,1 --reps 1 > $O/probe$p.log 2>&1
  python tools/trace_summary.py $O/trace$p.txt > $O/summary$p.txt 2>&1
  python tools/chain_evidence.py $O/trace$p.txt 10 > $O/chains$p.txt 2>&1
  python - $O/trace$p.txt > $O/late$p.txt <<'PY'
import sys, numpy as np
a = np.loadtxt(sys.argv[1], dtype=np.int64, ndmin=2)
t0 = a[:, 4].min(); st = (a[:, 4] - t0) / 1e5; en = (a[:, 5] - t0) / 1e5
late = np.argsort(-en)[:6]
for i in late:
    task = a[i, 0]
    m = a[:, 0] == task
    print(f"task {task}: " + "; ".join(f"beam {a[j,3]} sub {a[j,1]} big {a[j,2]} {st[j]:.2f}-{en[j]:.2f} found {a[j,6]}" for j in np.flatnonzero(m)[np.argsort(st[m])]))
PY
  rm -f $O/trace$p.txt
  grep "^2\^" $O/probe$p.log | cut -c1-200
  cat $O/late$p.txt
done
