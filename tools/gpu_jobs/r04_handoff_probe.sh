#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r04hprobe
mkdir -p $O
python tools/handoff_probe.py glove 30 > $O/glove_gate.log 2>&1
python tools/handoff_probe.py deep 40 > $O/deep_gate.log 2>&1
