#!/bin/bash
# round 5: raw parity tests, A/B probe (tree vs tools/_scratch/ab1 = first-generation core) and the phase profile in one call
export TMPDIR=/tmp
O=gpurun_out/r05mid
mkdir -p $O
rm -f $O/probe.log
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "raw_beam_search" > $O/raw_tests.log 2>&1
tail -5 $O/raw_tests.log
for v in tree ${AB:-tools/_scratch/ab1}; do
  if [ $v = tree ]; then L=$LD_LIBRARY_PATH; else L=$PWD/$v:$LD_LIBRARY_PATH; fi
  echo "== $v" >> $O/probe.log
  LD_LIBRARY_PATH=$L timeout 600 python tools/mid_core_probe.py 1000000 160,320,640,1280 64,8192 2>&1 | grep -v amdgpu.ids >> $O/probe.log
done
cat $O/probe.log
LD_LIBRARY_PATH=$PWD/tools/_scratch/prof:$LD_LIBRARY_PATH WANN_PROFILE_PHASES=1 timeout 600 python tools/mid_core_probe.py 1000000 160,1280 64,8192 2>&1 | grep -v "amdgpu.ids" > $O/prof.log
cat $O/prof.log
