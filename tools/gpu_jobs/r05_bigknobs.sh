#!/bin/bash
# round 5: one-wave core experiments (same results by construction): request look-ahead 8, filter store held behind both probes
export TMPDIR=/tmp
O=gpurun_out/r05bigknobs
mkdir -p $O
for v in ${VARIANTS:-tree tools/_scratch/la8 tools/_scratch/hold tools/_scratch/la8hold}; do
  if [ $v = tree ]; then L=$LD_LIBRARY_PATH; else L=$PWD/$v:$LD_LIBRARY_PATH; fi
  echo "== $v"
  LD_LIBRARY_PATH=$L python tools/phase_profile.py 5120,2560 1 2>&1 | grep "wann raw"
  LD_LIBRARY_PATH=$L timeout 600 python tools/frac_probe.py --fractions=${FRACS:--8,-9,-11} --settings 80,1 --reps 4 2>&1 | grep "^2\^" | sed -e 's/rounds.*big_searches/big_searches/' | cut -c1-220
done 2>&1 | tee $O/bigknobs.log
