#!/bin/bash
# round 5, last tree: the GPU suite as the driver runs it, smoke, then repetition stress
export TMPDIR=/tmp
O=gpurun_out/r05val
mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q > $O/gpu_tests_final.log 2>&1
tail -3 $O/gpu_tests_final.log | head -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-120
timeout 500 python tools/stress_repeat.py 420 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-200 | tee $O/stress_repeat.log
timeout 300 python tools/stress_prefilter.py 200 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-200 | tee $O/stress_prefilter.log
