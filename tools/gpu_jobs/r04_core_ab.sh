#!/bin/bash
# parity of the one-wave core on the tree's build, then the same-box A/B against tools/_scratch/ab*
export TMPDIR=/tmp
O=gpurun_out/r04ab
mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "raw or mid_fraction or lookahead or deep_chains or unserved or serialised or spec or index_matches or golden" > $O/tests.log 2>&1
bash tools/gpu_jobs/r04_ab.sh
