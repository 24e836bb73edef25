#!/bin/bash
O=gpurun_out/full
mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
WANN_PF_NO_REF=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 tools/bench_prefilter.py > $O/bench.json 2> $O/bench.err
