#!/bin/bash
O=gpurun_out/prefilter
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
WANN_PF_NO_REF=1 WANN_PF_DIM=512 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof512 -- python3 tools/bench_prefilter.py > $O/b512.json 2> $O/b512.err
