#!/bin/bash
# round 5: phase profile of the third-generation core (PROFILE=1 build in tools/_scratch/prof)
export TMPDIR=/tmp
O=gpurun_out/r05mid
mkdir -p $O
LD_LIBRARY_PATH=$PWD/tools/_scratch/prof:$LD_LIBRARY_PATH WANN_PROFILE_PHASES=1 timeout 600 python tools/mid_core_probe.py 1000000 ${BEAMS:-1280} 64,8192 > $O/prof.log 2>&1
grep -v amdgpu.ids $O/prof.log | tail -40
