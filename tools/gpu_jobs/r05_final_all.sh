#!/bin/bash
# round 5, final tree: the GPU suite as the driver runs it, then the profiler passes for profiles/
export TMPDIR=/tmp
O=gpurun_out/r05val
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests_final.log 2>&1
tail -4 $O/gpu_tests_final.log | head -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/gpu_jobs/r05_final.sh
