#!/bin/bash
# round 5: the default bench line (all fractions, configs, cpu baseline)
export TMPDIR=/tmp
O=gpurun_out/r05bench
mkdir -p $O
python bench.py --steps 20 --warmup 3 > $O/bench_n1.json 2> $O/bench_n1.log
tail -c 3000 $O/bench_n1.json
