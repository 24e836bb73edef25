#!/bin/bash
# the whole GPU suite as the driver runs it (incl. the full-size parity tests against the real reference)
export TMPDIR=/tmp
O=gpurun_out/r04tests
mkdir -p $O
( time python -m pytest tests -x -q -m gpu -rs --durations=12 ) > $O/tests.log 2>&1
