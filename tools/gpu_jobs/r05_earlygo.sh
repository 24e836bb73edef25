#!/bin/bash
# round 5 (late): a four-wave chain goes to a poller when its look-ahead is REQUESTED (one level earlier); build tools/_scratch/early (-DWANN_AB=11)
export TMPDIR=/tmp
O=gpurun_out/r05earlygo
mkdir -p $O
LD_LIBRARY_PATH=$PWD/tools/_scratch/early:$LD_LIBRARY_PATH timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "mid_fraction or lookahead or deep_chains or final_research" 2>&1 | tail -2
for v in tree tools/_scratch/early tree tools/_scratch/early; do
  if [ $v = tree ]; then L=$LD_LIBRARY_PATH; else L=$PWD/$v:$LD_LIBRARY_PATH; fi
  echo "== $v"
  LD_LIBRARY_PATH=$L timeout 900 python tools/frac_probe.py --fractions=${FRACS:--4,-5,-6,-7,-8,-9,-10,-11} --settings 80,1 --reps 4 2>&1 | grep "^2\^" | sed -e 's/rounds.*handoffs/handoffs/' -e 's/packet_hops.*//' | cut -c1-170
done 2>&1 | tee $O/earlygo.log
