#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r04tracecfg
mkdir -p $O
for c in glove deep; do
  s=40,1; [ $c = deep ] && s=80,1
  python tools/bench_configs.py --config $c --threads '' --cache /tmp/wann_cfg_cache --setting $s > $O/$c.plain.json 2> $O/$c.plain.log
  LD_LIBRARY_PATH=$PWD/tools/_scratch/trace:$LD_LIBRARY_PATH WANN_TASK_TRACE=$O/trace_$c.txt python tools/bench_configs.py --config $c --threads '' --cache /tmp/wann_cfg_cache --setting $s > $O/$c.trace.json 2> $O/$c.trace.log
  python tools/trace_summary.py $O/trace_$c.txt > $O/summary_$c.txt 2>&1
  rm -f $O/trace_$c.txt
done
LD_LIBRARY_PATH=$PWD/tools/_scratch/trace:$LD_LIBRARY_PATH WANN_TASK_TRACE=$O/trace_sift.txt python tools/frac_probe.py --fractions=-3 --settings 80,1 --reps 1 > $O/sift.log 2>&1
python tools/trace_summary.py $O/trace_sift.txt > $O/summary_sift.txt 2>&1
rm -f $O/trace_sift.txt
