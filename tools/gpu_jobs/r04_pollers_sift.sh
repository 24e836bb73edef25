#!/bin/bash
# the number of deep-chain pollers (a CU each) on the SIFT-1M fractions that use them, and on the byte-row kernels
export TMPDIR=/tmp
O=gpurun_out/r04poll
mkdir -p $O
: > $O/sift.log
for p in 4 8 16 4 16; do
  echo "== WANN_DEEP_POLLERS=$p" >> $O/sift.log
  WANN_DEEP_POLLERS=$p python tools/frac_probe.py --fractions=-1,-2,-3,-4,-5 --settings 80,1 --reps 4 2>&1 | grep "^2\^" | cut -c1-90 >> $O/sift.log
done
