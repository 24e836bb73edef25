#!/bin/bash
# round 6: in-cap continuations of a COMPANION-mode launch handed to idle pollers (dev A/B builds tools/_scratch/ab7: from beam 640,
# ab8: from beam 1 280; make EXTRA=-DWANN_AB=7|8), against the tree; parity of the mid-fraction tests under ab7
export TMPDIR=/tmp
O=gpurun_out/r06f
mkdir -p $O
BASE_LD=$LD_LIBRARY_PATH
for v in tree ab7 ab8 tree ab7 ab8; do
  if [ $v = tree ]; then export LD_LIBRARY_PATH=$BASE_LD; else export LD_LIBRARY_PATH=$PWD/tools/_scratch/$v:$BASE_LD; fi
  echo "== $v" >> $O/ab.log
  timeout 900 python tools/frac_probe.py --fractions=-3,-4,-5,-6,-7,-8,-9,-10,-11 --settings 80,1 --reps 4 2>&1 | grep "^2\^" | cut -c1-200 >> $O/ab.log
done
export LD_LIBRARY_PATH=$BASE_LD
cut -c1-150 $O/ab.log
LD_LIBRARY_PATH=$PWD/tools/_scratch/ab7:$BASE_LD timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "mid_fraction or final_research or scheduling_variants or lookahead or deep_chains or unserved or big_workgroup" > $O/tests_ab7.log 2>&1
tail -3 $O/tests_ab7.log
