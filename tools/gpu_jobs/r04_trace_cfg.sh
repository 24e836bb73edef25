#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r04tracecfg
mkdir -p $O
for c in deep; do
  s=80,1
  LD_LIBRARY_PATH=$PWD/tools/_scratch/trace:$LD_LIBRARY_PATH WANN_TASK_TRACE=$O/trace_$c.txt python tools/bench_configs.py --config $c --threads '' --cache /tmp/wann_cfg_cache --setting $s --seconds 2 > $O/$c.trace.json 2> $O/$c.trace.log
  python tools/trace_summary.py $O/trace_$c.txt > $O/summary_${c}_16pollers.txt 2>&1
  python tools/chain_evidence.py $O/trace_$c.txt 12 > $O/chains_${c}_16pollers.txt 2>&1
  rm -f $O/trace_$c.txt
done
