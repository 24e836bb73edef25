#!/bin/bash
# round 5: the "slot written" bitmap + second-pass row touches (tree) against the build without the bitmap (tools/_scratch/touch:
# touches only) and the first-generation core (tools/_scratch/ab1): parity, probe, fractions
export TMPDIR=/tmp
O=gpurun_out/r05wbits
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "raw_beam_search or mid_fraction or index_matches or golden" > $O/tests.log 2>&1
tail -3 $O/tests.log
rm -f $O/probe.log $O/mid.log
for v in tree tools/_scratch/touch tools/_scratch/ab1; do
  if [ $v = tree ]; then L=$LD_LIBRARY_PATH; else L=$PWD/$v:$LD_LIBRARY_PATH; fi
  echo "== $v" >> $O/probe.log
  LD_LIBRARY_PATH=$L timeout 600 python tools/mid_core_probe.py 1000000 160,320,640,1280 64,1024,8192 2>&1 | grep -v amdgpu.ids >> $O/probe.log
done
cat $O/probe.log
for v in tree tools/_scratch/touch; do
  if [ $v = tree ]; then L=$LD_LIBRARY_PATH; else L=$PWD/$v:$LD_LIBRARY_PATH; fi
  echo "== $v" >> $O/mid.log
  LD_LIBRARY_PATH=$L timeout 1200 python tools/frac_probe.py --fractions=-3,-5,-6,-7,-8,-9,-10,-11 --settings 80,1 --reps 4 2>&1 | grep "^2\^" | cut -c1-110 >> $O/mid.log
done
cat $O/mid.log
