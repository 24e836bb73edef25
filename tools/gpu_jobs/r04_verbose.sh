#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r04verbose
mkdir -p $O
python -m pytest tests/test_verbose.py tests/test_gpu_parity.py -x -q -m gpu -k "verbose or golden_reference or index_matches or raw_beam" > $O/tests.log 2>&1
