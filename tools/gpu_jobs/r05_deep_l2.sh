#!/bin/bash
# round 5: configs[3] under squared L2 (BASELINE.json's text): the n = 10^6 parity test of the suite, then the full-size leg with the
# real reference beside it (tools/bench_configs.py --config deep_l2)
export TMPDIR=/tmp
O=gpurun_out/r05deepl2
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -s -k "deep_like_l2" > $O/test_1m.log 2>&1
grep -E "fullsize|passed|failed|skipped" $O/test_1m.log | tail -6
timeout 1500 python tools/bench_configs.py --config deep_l2 --threads 32 --seconds 8 --cache /tmp/wann_cfg_cache > $O/deep_l2.json 2> $O/deep_l2.log
tail -5 $O/deep_l2.log | cut -c1-300
head -c 1500 $O/deep_l2.json
