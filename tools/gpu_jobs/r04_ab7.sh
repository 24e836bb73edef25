#!/bin/bash
# A/B 7: the squared-L2 float four-wave kernel at three waves per SIMD (one row per lane pair and pass, 163 registers, lean LDS pool)
export TMPDIR=/tmp
O=gpurun_out/r04ab7
mkdir -p $O
LD_LIBRARY_PATH=$PWD/tools/_scratch/ab7:$LD_LIBRARY_PATH python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "raw_beam or index_matches or golden_reference or mid_fraction" > $O/tests.log 2>&1
: > $O/sift.log
for v in main ab7 main ab7; do
  L=$LD_LIBRARY_PATH; [ $v = ab7 ] && L=$PWD/tools/_scratch/ab7:$LD_LIBRARY_PATH
  echo "== $v" >> $O/sift.log
  LD_LIBRARY_PATH=$L python tools/frac_probe.py --fractions=-1,-2,-3,-4,-5,-6,-7,-9 --settings 80,1 --reps 4 2>&1 | grep "^2\^" | cut -c1-90 >> $O/sift.log
done
