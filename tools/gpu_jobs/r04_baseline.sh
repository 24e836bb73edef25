#!/bin/bash
# round-4 baseline on this round's box: GPU tests, phase profile of the one-wave core at long beams, headline + mid fractions
export TMPDIR=/tmp
O=gpurun_out/r04base
mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1
LD_LIBRARY_PATH=$PWD/tools/_scratch/prof:$LD_LIBRARY_PATH python tools/phase_profile.py 5120,2560,1280 1,4 > $O/phases.log 2>&1
python tools/phase_profile.py 5120,2560,1280 1,4 > $O/phases_plain.log 2>&1
python bench.py --fractions headline --configs none --no-cpu-baseline --setting 80,1 --steps 20 --warmup 3 > $O/head.json 2> $O/head.log
python tools/frac_probe.py --fractions=-6,-8,-9,-11 --settings 80,1 --reps 3 > $O/mid.log 2>&1
