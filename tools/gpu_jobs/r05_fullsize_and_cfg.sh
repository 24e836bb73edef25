#!/bin/bash
# round 5: the full-size parity file (new mid-fraction legs), then the GloVe-like / deep-like legs' timing (no reference threads)
export TMPDIR=/tmp
O=gpurun_out/r05cfg
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -x -q -s > $O/fullsize.log 2>&1
grep "fullsize\]\|passed\|failed\|Error" $O/fullsize.log | cut -c1-260
for c in glove deep; do
  s=40,1; [ $c = deep ] && s=80,1
  timeout 900 python tools/bench_configs.py --config $c --threads '' --cache /tmp/wann_fullsize_cache/cfg --setting $s > $O/$c.json 2> $O/$c.log
  python3 - <<PY
import json
d=json.load(open("$O/$c.json"))
print("$c", {k: d[k] for k in d if k in ("qps","device_ms","search_kernel_ms","roofline","setting","ms_per_batch","search_kernel_ms_per_call")})
PY
done
