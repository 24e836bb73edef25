#!/bin/bash
# same-box A/B of an environment switch on the throughput-bound legs
export TMPDIR=/tmp
O=gpurun_out/r04abenv
mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "index_matches or golden_reference or mid_fraction or edge or fenwick" > $O/tests.log 2>&1
: > $O/ab.log
for v in A=1 WANN_NO_ORDER_PLAIN=1 A=2 WANN_NO_ORDER_PLAIN=1; do
  echo "== $v" >> $O/ab.log
  env $v python bench.py --fractions headline --configs none --no-cpu-baseline --setting 80,1 --steps 20 --warmup 3 --pipeline 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sift', d['value'], d['roofline']['kernel_ms_per_step'], d['roofline']['frac'])" >> $O/ab.log
  for c in glove deep; do
    s=40,1; [ $c = deep ] && s=80,1
    env $v python tools/bench_configs.py --config $c --threads '' --cache /tmp/wann_cfg_cache --setting $s 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config'], d['setting'], d['ms_per_batch'], d['search_kernel_ms'], d.get('k_search_tb_per_s'))" >> $O/ab.log
  done
  env $v python tools/frac_probe.py --fractions=-2,-4,-5 --settings 80,1 --reps 4 2>&1 | grep "^2\^" | cut -c1-75 >> $O/ab.log
done
