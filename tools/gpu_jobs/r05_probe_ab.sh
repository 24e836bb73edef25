#!/bin/bash
# round 5: mid_core_probe on the tree's build and on the builds listed in AB (same box)
export TMPDIR=/tmp
O=gpurun_out/r05mid
mkdir -p $O
rm -f $O/probe_ab.log
for v in tree ${AB}; do
  if [ $v = tree ]; then L=$LD_LIBRARY_PATH; else L=$PWD/$v:$LD_LIBRARY_PATH; fi
  echo "== $v" >> $O/probe_ab.log
  LD_LIBRARY_PATH=$L timeout 600 python tools/mid_core_probe.py ${ARGS:-1000000 640,1280 64,1024,2048,4096,8192} 2>&1 | grep -v amdgpu.ids >> $O/probe_ab.log
done
cat $O/probe_ab.log
