#!/bin/bash
# same-box A/B of kernel experiments: tools/_scratch/ab*/libwann.so against the tree's build (isolated long searches, then mid fractions)
export TMPDIR=/tmp
O=gpurun_out/r04ab
mkdir -p $O
: > $O/raw.log
for rep in 1 2 3; do
  for v in tree $(ls -d tools/_scratch/ab* 2>/dev/null); do
    if [ $v = tree ]; then L=$LD_LIBRARY_PATH; else L=$PWD/$v:$LD_LIBRARY_PATH; fi
    echo "== $v rep $rep" >> $O/raw.log
    LD_LIBRARY_PATH=$L python tools/phase_profile.py 5120,2560 1 2>&1 | grep "wann raw" >> $O/raw.log
  done
done
for v in tree $(ls -d tools/_scratch/ab* 2>/dev/null); do
  if [ $v = tree ]; then L=$LD_LIBRARY_PATH; else L=$PWD/$v:$LD_LIBRARY_PATH; fi
  echo "== $v" >> $O/mid.log
  LD_LIBRARY_PATH=$L python tools/frac_probe.py --fractions=-5,-6,-7,-8,-9,-10,-11 --settings 80,1 --reps 4 2>&1 | grep "^2\^" | cut -c1-75 >> $O/mid.log
done
