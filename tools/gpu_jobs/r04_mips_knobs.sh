#!/bin/bash
# deep-10M-like leg: three workgroups per CU (lean pool) x deep-chain pollers, same box (the graph cache is built once)
export TMPDIR=/tmp
O=gpurun_out/r04mips
mkdir -p $O
: > $O/ab2.log
run() {  # config setting env...
  c=$1; s=$2; shift 2
  echo "== $c $s $*" >> $O/ab2.log
  env "$@" python tools/bench_configs.py --config $c --threads '' --cache /tmp/wann_cfg_cache --setting $s --seconds 4 2> $O/err.tmp | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config'], d['setting'], d['ms_per_batch'], d['search_kernel_ms'], d.get('k_search_tb_per_s'))" >> $O/ab2.log
}
for i in 1 2; do
run deep 80,1 A=$i
run deep 80,1 WANN_NO_LEAN=1 WANN_DEEP_POLLERS=4
run deep 80,1 WANN_NO_LEAN=1 WANN_DEEP_POLLERS=16
run deep 80,1 WANN_DEEP_POLLERS=4
run deep 80,1 WANN_DEEP_POLLERS=24
done
