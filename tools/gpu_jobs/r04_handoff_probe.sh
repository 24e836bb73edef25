#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r04hprobe
mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "deep_chains or serialised or unserved or index_matches or mid_fraction" > $O/tests.log 2>&1
python tools/handoff_probe.py sift 30 > $O/sift_gate.log 2>&1
WANN_NO_GATE=1 python tools/handoff_probe.py sift 30 > $O/sift_nogate.log 2>&1
python tools/handoff_probe.py sift 30 > $O/sift_gate2.log 2>&1
python bench.py --fractions headline --configs none --no-cpu-baseline --steps 40 --warmup 3 > $O/head.json 2> $O/head.log
WANN_NO_GATE=1 python bench.py --fractions headline --configs none --no-cpu-baseline --steps 40 --warmup 3 > $O/head_nogate.json 2> $O/head_nogate.log
