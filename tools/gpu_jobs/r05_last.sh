#!/bin/bash
# round 5, last code change (inner-product kernels: compile-time distance routines in the general core, no row touches there):
# the GPU suite, the inner-product probe against the first-generation core, the GloVe-like / deep-like legs, the mid fractions
export TMPDIR=/tmp
O=gpurun_out/r05last
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1
tail -4 $O/gpu_tests.log | head -2
AB="tools/_scratch/ab1" ARGS="1000000 160,320,640 64,8192 1 96" bash tools/gpu_jobs/r05_probe_ab.sh > $O/probe_mips.log 2>&1
cat $O/probe_mips.log
for c in glove deep; do
  s=40,1; [ $c = deep ] && s=80,1
  timeout 900 python tools/bench_configs.py --config $c --threads '' --cache /tmp/wann_fullsize_cache/cfg --setting $s > $O/$c.json 2> $O/$c.log
  python3 - <<PY
import json
d=json.load(open("$O/$c.json"))
print("$c", {k: d[k] for k in d if k in ("qps","search_kernel_ms","ms_per_batch","k_search_tb_per_s","pipelined")})
PY
done
VARIANTS=tree bash tools/gpu_jobs/r05_frac_ab.sh
