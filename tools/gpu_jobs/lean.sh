#!/bin/bash
O=gpurun_out/lean.log
: > $O
run() { echo "== $*" >> $O; env "$@" python tools/bench_configs.py --config deep --threads '' --setting 80,1 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_batch'], d['search_kernel_ms'], d.get('k_search_tb_per_s'))" >> $O; }
run A=1
run WANN_NO_LEAN=1
run WANN_LEAN_POOL=11776
run WANN_LEAN_POOL=12224
run WANN_LEAN_POOL=11776 WANN_NO_DEEP=1
