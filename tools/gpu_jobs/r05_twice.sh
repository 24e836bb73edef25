#!/bin/bash
# round 5: the exact (per-hop multiset) union only for rows that list a node twice, not for every row with a shared filter slot:
# raw parity (the core-variant test has such rows), then probe and fractions against the previous commit's build (tools/_scratch/prev)
export TMPDIR=/tmp
O=gpurun_out/r05twice
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "raw_beam_search or mid_fraction or index_matches or golden" > $O/tests.log 2>&1
tail -3 $O/tests.log
AB="tools/_scratch/prev" ARGS="1000000 160,320,640 64,8192" bash tools/gpu_jobs/r05_probe_ab.sh > $O/probe.log 2>&1
cat $O/probe.log
VARIANTS="tree tools/_scratch/prev" FRACS=-3,-4,-5,-6,-7 bash tools/gpu_jobs/r05_frac_ab.sh
