#!/bin/bash
# round 5: the one-wave core leaves a filter slot alone when it already holds the id (fewer dirty lines of a table of up to 32 MiB)
export TMPDIR=/tmp
O=gpurun_out/r05big
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "raw_beam_search or mid_fraction or deep_chains or lookahead or final_research" > $O/tests.log 2>&1
tail -3 $O/tests.log
for v in tree tools/_scratch/prev; do
  if [ $v = tree ]; then L=$LD_LIBRARY_PATH; else L=$PWD/$v:$LD_LIBRARY_PATH; fi
  echo "== $v"
  LD_LIBRARY_PATH=$L python tools/phase_profile.py 5120,2560 1,4 2>&1 | grep "wann raw"
done
VARIANTS="tree tools/_scratch/prev" FRACS=-6,-8,-9,-10,-11 bash tools/gpu_jobs/r05_frac_ab.sh
