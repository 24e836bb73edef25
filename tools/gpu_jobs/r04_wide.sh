#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r04wide
mkdir -p $O
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "wide_row_index" > $O/tests.log 2>&1
python -m pytest tests/test_vamana_api.py -x -q -m gpu >> $O/tests.log 2>&1
