#!/bin/bash
# round 5: the one-wave core probes the filter two hops ahead (packets carry partner lanes): parity subset, lone chains, fractions A/B
export TMPDIR=/tmp
O=gpurun_out/r05early
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "raw_beam_search or mid_fraction or deep_chains or lookahead or final_research" > $O/tests.log 2>&1
tail -3 $O/tests.log
for v in tree tools/_scratch/prev; do
  if [ $v = tree ]; then L=$LD_LIBRARY_PATH; else L=$PWD/$v:$LD_LIBRARY_PATH; fi
  echo "== $v"
  LD_LIBRARY_PATH=$L python tools/phase_profile.py 5120,2560 1,4 2>&1 | grep "wann raw"
done
VARIANTS="tree tools/_scratch/prev" FRACS=${FRACS:--6,-8,-9,-10,-11} bash tools/gpu_jobs/r05_frac_ab.sh
