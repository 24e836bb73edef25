#!/bin/bash
# scheduling knobs at the mid window fractions (same box, same process image): ms per 10 000-query batch
O=gpurun_out/knobs.log
: > $O
run() { echo "== $*" >> $O; env "$@" python tools/frac_probe.py --fractions=-8,-9,-11,-7,-6 --settings 80,1 --reps 3 2>&1 | grep "^2\^" | cut -c1-75 >> $O; }
run A=1
run WANN_SCAN=1
run WANN_BIG_EXCLUSIVE=1
run WANN_BIG_EXCLUSIVE=1 WANN_SCAN=1
run WANN_SCAN=1 WANN_POLLERS=64
