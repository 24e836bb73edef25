#!/bin/bash
O=gpurun_out/prefilter
mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "prefilter or dense or gemm" > $O/tests.log 2>&1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
WANN_PF_ONLY=p12 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_p12 -- python3 tools/bench_prefilter.py > $O/p12.json 2> $O/p12.err
WANN_PF_NO_REF=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 tools/bench_prefilter.py > $O/bench.json 2> $O/bench.err
