#!/bin/bash
# round 5: same-box A/B on the bench index (SIFT-1M-like 2-WST): tree vs tools/_scratch/ab1 (first-generation core in the four-wave kernel)
export TMPDIR=/tmp
O=gpurun_out/r05frac
mkdir -p $O
rm -f $O/mid.log
for v in ${VARIANTS:-tree tools/_scratch/ab1}; do
  if [ $v = tree ]; then L=$LD_LIBRARY_PATH; else L=$PWD/$v:$LD_LIBRARY_PATH; fi
  echo "== $v" >> $O/mid.log
  LD_LIBRARY_PATH=$L timeout 1200 python tools/frac_probe.py --fractions=${FRACS:--3,-5,-6,-7,-8,-9,-10,-11} --settings 80,1 --reps 4 2>&1 | grep "^2\^" | cut -c1-150 >> $O/mid.log
done
cat $O/mid.log
