#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r04async
mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "asynchronous" > $O/tests.log 2>&1
: > $O/ab.log
for v in A=1 WANN_NO_ASYNC_ROOM=1 A=2 WANN_NO_ASYNC_ROOM=1; do
  echo "== $v" >> $O/ab.log
  env $v python bench.py --fractions headline --configs none --no-cpu-baseline --setting 80,1 --steps 40 --warmup 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['frac'], d['config']['pipelined'])" >> $O/ab.log
done
