#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r04hprobe
mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "deep_chains or serialised or unserved" > $O/tests3.log 2>&1
python tools/handoff_probe.py sift 30 > $O/sift_gate3.log 2>&1
