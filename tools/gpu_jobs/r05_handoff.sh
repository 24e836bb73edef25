#!/bin/bash
# round 5 (late): chains of the four-wave kernel hand their next level to an idle poller of the companion launch also in companion mode
export TMPDIR=/tmp
O=gpurun_out/r05handoff
mkdir -p $O
for h in ${HANDOFFS:-0 320 640 1280}; do
  echo "== WANN_HANDOFF_COMPANION=$h"
  WANN_HANDOFF_COMPANION=$h timeout 900 python tools/frac_probe.py --fractions=${FRACS:--5,-6,-7,-8,-9,-10,-11} --settings 80,1 --reps ${REPS:-4} 2>&1 | grep "^2\^" | sed -e 's/rounds.*handoffs/handoffs/' -e 's/packet_hops.*//' | cut -c1-200
done 2>&1 | tee $O/handoff.log
