#!/bin/bash
# timelines of the fractions whose batch time jumps between two values (2^-6, 2^-7, 2^-8, 2^-11): default setting and WANN_SCAN_NUM=8
export TMPDIR=/tmp
O=gpurun_out/traces2
mkdir -p $O
export LD_LIBRARY_PATH=$PWD/tools/_scratch/trace:$LD_LIBRARY_PATH
for v in dflt num8; do
  E="A=1"; [ $v = num8 ] && E="WANN_SCAN_NUM=8"
  for p in -6 -7 -8 -11; do
    env $E WANN_TASK_TRACE=$O/trace$p.txt python tools/frac_probe.py --fractions=$p --settings 80,1 --reps 1 > $O/probe${p}_$v.log 2>&1
    python tools/trace_summary.py $O/trace$p.txt > $O/summary${p}_$v.txt 2>&1
    python tools/chain_evidence.py $O/trace$p.txt 8 > $O/chains${p}_$v.txt 2>&1
    rm -f $O/trace$p.txt
  done
done
