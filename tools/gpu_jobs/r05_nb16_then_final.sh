#!/bin/bash
# the compiled-in block count (d = 128): raw parity + probe; then the round's profile passes
export TMPDIR=/tmp
O=gpurun_out/r05mid
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "raw_beam_search or mid_fraction or index_matches" > $O/raw_tests2.log 2>&1
tail -3 $O/raw_tests2.log
timeout 600 python tools/mid_core_probe.py 1000000 160,320,640,1280 64,8192 2>&1 | grep -v amdgpu.ids | tee $O/probe_nb16.log
bash tools/gpu_jobs/r05_final.sh
