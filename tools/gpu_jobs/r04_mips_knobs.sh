#!/bin/bash
# launch shapes and occupancy knobs of the GloVe-like / deep-like legs (one box: the graph cache is built once)
export TMPDIR=/tmp
O=gpurun_out/r04mips
mkdir -p $O
: > $O/ab.log
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "deep_chains or index_matches or mid_fraction or unserved" > $O/tests.log 2>&1
run() {  # config setting env...
  c=$1; s=$2; shift 2
  echo "== $c $s $*" >> $O/ab.log
  env "$@" python tools/bench_configs.py --config $c --threads '' --cache /tmp/wann_cfg_cache --setting $s --seconds 4 2> $O/err.tmp | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config'], d['setting'], d['ms_per_batch'], d['search_kernel_ms'], d.get('k_search_tb_per_s'))" >> $O/ab.log
}
for i in 1 2; do
run glove 40,1 A=1
run glove 40,1 WANN_DEEP_MIN_TASKS=2000
run glove 40,1 WANN_DEEP_MIN_TASKS=2000 WANN_DEEP_POLLERS=8
run glove 40,1 WANN_DEEP_MIN_TASKS=2000 WANN_NO_LATE_HANDOFF=1
done
for i in 1 2; do
run deep 80,1 A=1
run deep 80,1 WANN_NO_LATE_HANDOFF=1
run deep 80,1 WANN_DEEP_POLLERS=32
done
