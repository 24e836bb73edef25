#!/bin/bash
# round 5: the mid fractions are bound by the companion launch's beam-5 120 chains now: do exclusive CUs for them pay since the bulk got faster?
export TMPDIR=/tmp
O=gpurun_out/r05knobs
mkdir -p $O
rm -f $O/knobs.log
for e in "X=0" "WANN_BIG_EXCLUSIVE=1" "WANN_POLLERS=64" "WANN_SPEC_NUM=6"; do
  echo "== $e" >> $O/knobs.log
  env $e timeout 900 python tools/frac_probe.py --fractions=-6,-7,-8,-9,-11 --settings 80,1 --reps 4 2>&1 | grep "^2\^" | cut -c1-110 >> $O/knobs.log
done
cat $O/knobs.log
