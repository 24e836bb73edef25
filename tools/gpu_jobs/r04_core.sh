#!/bin/bash
# round-4 core iteration: parity of the one-wave core, its phase profile, isolated long searches, the mid window fractions
export TMPDIR=/tmp
O=gpurun_out/r04core
mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "raw or mid_fraction or lookahead or deep_chains or unserved or serialised" > $O/tests.log 2>&1
LD_LIBRARY_PATH=$PWD/tools/_scratch/prof:$LD_LIBRARY_PATH python tools/phase_profile.py 5120,2560 1 > $O/phases.log 2>&1
python tools/phase_profile.py 5120,2560 1,4 > $O/phases_plain.log 2>&1
python tools/frac_probe.py --fractions=-6,-8,-9,-11 --settings 80,1 --reps 3 > $O/mid.log 2>&1
