#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r04small
mkdir -p $O
LD_LIBRARY_PATH=$PWD/tools/_scratch/prof:$LD_LIBRARY_PATH python tools/phase_profile_small.py 1000000 80 64 1:96 > $O/phases_hit.log 2>&1
