#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r04final3
mkdir -p $O
python bench.py --steps 20 --warmup 3 > $O/bench_n1.json 2> $O/bench_n1.log
