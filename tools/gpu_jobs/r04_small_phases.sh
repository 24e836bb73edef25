#!/bin/bash
# per-phase cycles of the small-beam core (PROFILE build of the shipped code) on stand-alone graphs, idle machine and under load
export TMPDIR=/tmp
O=gpurun_out/r04small
mkdir -p $O
LD_LIBRARY_PATH=$PWD/tools/_scratch/prof:$LD_LIBRARY_PATH python tools/phase_profile_small.py 1000000 40,80,160 64,10000 > $O/phases_shipped.log 2>&1
