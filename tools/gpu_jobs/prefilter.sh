#!/bin/bash
O=gpurun_out/prefilter
mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "prefilter or dense or gemm" > $O/tests.log 2>&1
WANN_PF_NO_REF=1 python tools/bench_prefilter.py > $O/bench.json 2> $O/bench.err
