#!/bin/bash
# final state of round 4: the whole GPU suite, then the default bench line
export TMPDIR=/tmp
O=gpurun_out/r04final4
mkdir -p $O
( time python -m pytest tests -x -q -m gpu -rs --durations=8 ) > $O/tests.log 2>&1
python bench.py --steps 20 --warmup 3 > $O/bench_n1.json 2> $O/bench_n1.log
