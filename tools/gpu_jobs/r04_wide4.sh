#!/bin/bash
# the four-slab dense-prefilter kernel with the fetch overlapped (k_gemm_scores_wide4) against the round-3 kernel (WANN_AB=8 build)
export TMPDIR=/tmp
O=gpurun_out/r04wide4
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "dense_prefilter" > $O/tests.log 2>&1
export WANN_PF_NO_REF=1
: > $O/summary.txt
for v in new old new old; do
  L=$LD_LIBRARY_PATH; [ $v = old ] && L=$PWD/tools/_scratch/ab8:$LD_LIBRARY_PATH
  LD_LIBRARY_PATH=$L WANN_PF_DIM=512 timeout 600 python tools/bench_prefilter.py > $O/d512_$v.json 2> $O/d512_$v.log
  echo "== $v" >> $O/summary.txt
  python -c "
import json,sys
j=json.loads([l for l in open('$O/d512_$v.json') if l.startswith('{')][-1])
print({k:j[k] for k in j if 'ms' in k or 'gemm' in k.lower()})" >> $O/summary.txt 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 tools/bench_prefilter.py > $O/kt.json 2> $O/kt.log
find $O -name '*kernel_trace.csv' -delete
