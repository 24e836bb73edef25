#!/bin/bash
# launch shapes and occupancy knobs of the GloVe-like / deep-like legs (one box: the graph cache is built once)
export TMPDIR=/tmp
O=gpurun_out/r04mips
mkdir -p $O
: > $O/ab.log
run() {  # config setting env...
  c=$1; s=$2; shift 2
  echo "== $c $s $*" >> $O/ab.log
  env "$@" python tools/bench_configs.py --config $c --threads '' --cache /tmp/wann_cfg_cache --setting $s --seconds 4 2> $O/err.tmp | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config'], d['setting'], d['ms_per_batch'], d['search_kernel_ms'], d.get('k_search_tb_per_s'))" >> $O/ab.log
  grep "wann launch" $O/err.tmp | sort | uniq -c | head -4 >> $O/ab.log
}
for c in deep; do
  s=80,1
  run $c $s A=1
  for p in 8 12 16 24 32; do run $c $s WANN_DEEP_POLLERS=$p; done
  run $c $s WANN_DEEP_POLLERS=4
  run $c $s WANN_DEEP_POLLERS=16 WANN_NO_LEAN=1
done
