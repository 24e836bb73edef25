#!/bin/bash
# round 5: the one-wave core finds the packets of forgotten requests by node (after a merge of the delta list): parity subset + A/B
export TMPDIR=/tmp
O=gpurun_out/r05lookup
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "raw_beam_search or mid_fraction or deep_chains or lookahead or final_research" > $O/tests.log 2>&1
tail -1 $O/tests.log
VARIANTS="${VARIANTS:-tree tools/_scratch/nolookup}" FRACS=-6,-8,-9,-10,-11 bash tools/gpu_jobs/r05_bigknobs.sh
