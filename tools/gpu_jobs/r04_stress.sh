#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r04stress
mkdir -p $O
python tools/stress_repeat.py 150 > $O/fixed.log 2>&1
if [ -d tools/_scratch/ab_head ]; then LD_LIBRARY_PATH=$PWD/tools/_scratch/ab_head:$LD_LIBRARY_PATH python tools/stress_repeat.py 90 > $O/before_fix.log 2>&1; fi
