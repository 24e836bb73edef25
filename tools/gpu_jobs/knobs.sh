#!/bin/bash
O=gpurun_out/knobs.log
: > $O
run() { echo "== $*" >> $O; env "$@" python tools/frac_probe.py --fractions=-6,-7,-8,-9,-10,-11 --settings 80,1 --reps 3 2>&1 | grep "^2\^" | cut -c1-75 >> $O; }
run A=1
run WANN_POLLERS=16
run WANN_POLLERS=48
run WANN_SCAN_NUM=12
run WANN_SCAN_NUM=24
run WANN_SCAN_MIN_TOP=1280
