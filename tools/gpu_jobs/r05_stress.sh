#!/bin/bash
# round 5, final code: repeated mid-fraction batches (rows and work counters of every repetition must equal the first one's) and
# random shared-window prefilter batches (dense MFMA path against the exact scan)
export TMPDIR=/tmp
O=gpurun_out/r05stress
mkdir -p $O
timeout 600 python tools/stress_repeat.py 300 2>&1 | grep -v amdgpu.ids | tail -5 | tee $O/stress_repeat.log
timeout 400 python tools/stress_prefilter.py 180 2>&1 | grep -v amdgpu.ids | tail -5 | tee $O/stress_prefilter.log
