#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r04mgpu
mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_distributed_gpu.py -x -q -m gpu -k "allgather or predicted or asynchronous or c_abi or sharded or launcher or multi_device" > $O/tests.log 2>&1
