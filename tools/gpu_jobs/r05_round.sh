#!/bin/bash
# round 5: raw parity tests + A/B probe + phase profile, then the mid-fraction A/B on the bench index, then the full-size SIFT test
export TMPDIR=/tmp
bash tools/gpu_jobs/r05_mid_all.sh 2>&1 | grep -v "phases 5"
bash tools/gpu_jobs/r05_frac_ab.sh
O=gpurun_out/r05cfg
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -s -k "sift" > $O/fullsize_sift.log 2>&1
grep "fullsize\]\|passed\|failed\|Error" $O/fullsize_sift.log | cut -c1-200
