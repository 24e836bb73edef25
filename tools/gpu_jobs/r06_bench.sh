#!/bin/bash
# round 6: the default bench line (all fractions, all configs incl. sift_u8 / fenwick / three_split, cpu baseline)
export TMPDIR=/tmp
O=gpurun_out/r06bench
mkdir -p $O
SECONDS=0; python bench.py --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.log; echo "bench.py wall: $SECONDS s"
tail -c 2500 $O/bench_n1.json
