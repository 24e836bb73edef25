#!/bin/bash
O=gpurun_out/knobs.log
: > $O
run() { echo "== $*" >> $O; env "$@" python tools/frac_probe.py --fractions=-9,-8 --settings 80,1 --reps 3 2>&1 | grep "^2\^" | cut -c1-75 >> $O; }
run A=1
run WANN_BLOCKS_PER_CU=1
run WANN_NO_HELPER=1
