#!/bin/bash
O=gpurun_out/traces
mkdir -p $O
export LD_LIBRARY_PATH=$PWD/tools/_scratch/trace:$LD_LIBRARY_PATH
for p in -9 -8 -10; do
  WANN_TASK_TRACE=$O/trace$p.txt python tools/frac_probe.py --fractions=$p --settings 80,1 --reps 1 > $O/probe$p.log 2>&1
  python tools/trace_summary.py $O/trace$p.txt > $O/summary$p.txt 2>&1
  python tools/chain_evidence.py $O/trace$p.txt 10 > $O/chains$p.txt 2>&1
  rm -f $O/trace$p.txt
done
