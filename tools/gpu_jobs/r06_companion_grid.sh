#!/bin/bash
# round 6: the companion launch sized by its items + pollers instead of one workgroup per CU: same-box A/B against tools/_scratch/head,
# parity of the mid-fraction tests, timeline at 2^-7
export TMPDIR=/tmp
O=gpurun_out/r06h
mkdir -p $O
BASE_LD=$LD_LIBRARY_PATH
for v in tree head tree head; do
  if [ $v = tree ]; then export LD_LIBRARY_PATH=$BASE_LD; else export LD_LIBRARY_PATH=$PWD/tools/_scratch/head:$BASE_LD; fi
  echo "== $v" >> $O/ab.log
  timeout 900 python tools/frac_probe.py --fractions=-3,-4,-5,-6,-7,-8,-9,-10,-11 --settings 80,1:160,1 --reps 4 2>&1 | grep "^2\^" | cut -c1-200 >> $O/ab.log
done
export LD_LIBRARY_PATH=$BASE_LD
cut -c1-120 $O/ab.log
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -x -k "mid_fraction or final_research or scheduling_variants or lookahead or deep_chains or unserved or big_workgroup or serialised or asynchronous or sift_1m" > $O/tests.log 2>&1
tail -3 $O/tests.log
timeout 300 python tools/stress_repeat.py 150 > $O/stress.log 2>&1; tail -1 $O/stress.log | cut -c1-200
for c in glove deep; do
  timeout 900 python tools/bench_configs.py --config $c --threads '' --cache /tmp/wann_cfg_cache > $O/$c.json 2> $O/$c.log
  python3 -c "
import json
d=json.loads([l for l in open('$O/$c.json') if l.startswith('{')][-1])
print('$c', {k:d[k] for k in ('setting','ms_per_batch','qps','search_kernel_ms','k_search_tb_per_s')}, d['pipelined'].get('speedup_over_blocking'))"
done
