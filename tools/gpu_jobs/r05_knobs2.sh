#!/bin/bash
# round 5: with the four-wave kernel faster, does the companion chain gain from fewer ordinary waves per CU or from issue priority?
export TMPDIR=/tmp
O=gpurun_out/r05knobs
mkdir -p $O
rm -f $O/knobs2.log
for e in "X=0" "WANN_BLOCKS_PER_CU=1" "WANN_SEARCH_PRIO=1"; do
  echo "== $e" >> $O/knobs2.log
  env $e timeout 900 python tools/frac_probe.py --fractions=-3,-6,-8,-9,-11 --settings 80,1 --reps 4 2>&1 | grep "^2\^" | cut -c1-110 >> $O/knobs2.log
done
cat $O/knobs2.log
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "per_query_ids" 2>&1 | tail -3
