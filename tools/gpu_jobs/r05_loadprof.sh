#!/bin/bash
# round 5: phase cycles of the companion launch's long chains UNDER the batch's load (make PROFILE=1 builds)
export TMPDIR=/tmp
O=gpurun_out/r05loadprof
mkdir -p $O
for v in ${VARIANTS:-tools/_scratch/prof_old tools/_scratch/prof_new}; do
  echo "== $v"
  LD_LIBRARY_PATH=$PWD/$v:$LD_LIBRARY_PATH WANN_PROFILE_PHASES=1 timeout 600 python tools/frac_probe.py --fractions=-9 --settings 80,1 --reps 2 2>&1 | grep -E "companion phases|^2\^" | cut -c1-700 | tail -3
  LD_LIBRARY_PATH=$PWD/$v:$LD_LIBRARY_PATH python tools/phase_profile.py 5120 1 2>&1 | grep -E "wann phases|wann raw"
done 2>&1 | tee $O/loadprof.log
