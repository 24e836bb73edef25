#!/bin/bash
# how long does a needed beam-5120 search take when its workgroup has the CU to itself?  (timeline with WANN_BIG_EXCLUSIVE=1)
export TMPDIR=/tmp
O=gpurun_out/r04excl
mkdir -p $O
export LD_LIBRARY_PATH=$PWD/tools/_scratch/trace:$LD_LIBRARY_PATH
for v in A=1 WANN_BIG_EXCLUSIVE=1; do
  env $v WANN_TASK_TRACE=$O/trace_$v.txt python tools/frac_probe.py --fractions=-9 --settings 80,1 --reps 1 > $O/probe_$v.log 2>&1
  python tools/trace_summary.py $O/trace_$v.txt > $O/summary_$v.txt 2>&1
  rm -f $O/trace_$v.txt
done
