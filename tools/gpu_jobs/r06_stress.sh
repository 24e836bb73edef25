#!/bin/bash
# round 6: repetition stress of the final tree -- multi-task methods with their scans aside, prefilter proof path
export TMPDIR=/tmp
O=gpurun_out/r06stress
mkdir -p $O
timeout 400 python tools/stress_methods.py 240 > $O/stress_methods.log 2>&1; tail -2 $O/stress_methods.log | cut -c1-300
timeout 400 python tools/stress_prefilter.py 180 > $O/stress_prefilter.log 2>&1; tail -2 $O/stress_prefilter.log | cut -c1-300
