#!/bin/bash
# round 5: the third-generation general core of the four-wave kernel -- raw parity tests, then same-box A/B against the
# first-generation core (tools/_scratch/ab1 = make EXTRA=-DWANN_AB=1)
export TMPDIR=/tmp
O=gpurun_out/r05mid
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "raw_beam_search" > $O/raw_tests.log 2>&1
tail -5 $O/raw_tests.log
for v in tree tools/_scratch/ab1; do
  if [ $v = tree ]; then L=$LD_LIBRARY_PATH; else L=$PWD/$v:$LD_LIBRARY_PATH; fi
  echo "== $v" >> $O/probe.log
  LD_LIBRARY_PATH=$L timeout 600 python tools/mid_core_probe.py 1000000 160,320,640,1280 64,8192 2>&1 | grep -v amdgpu.ids >> $O/probe.log
done
cat $O/probe.log
