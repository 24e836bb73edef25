#!/bin/bash
O=gpurun_out/r04knobs.log
: > $O
run() { echo "== $*" >> $O; env "$@" python tools/frac_probe.py --fractions=-7,-8,-9,-11 --settings 80,1 --reps 4 2>&1 | grep "^2\^" | cut -c1-75 >> $O; }
run A=1
run WANN_BIG_EXCLUSIVE=1
run WANN_POLLERS=64
run WANN_HEAVY_RATIO=4
run WANN_SPEC_NUM=6
run WANN_SPEC_NUM=12
