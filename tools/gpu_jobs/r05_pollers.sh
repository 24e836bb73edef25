#!/bin/bash
# round 5 (late): more pollers (companion workgroups that wait for chains and look-aheads) with and without companion-mode hand-offs
export TMPDIR=/tmp
O=gpurun_out/r05pollers
mkdir -p $O
for cfg in ${CFGS:-"0 0" "128 0" "256 0" "128 640" "256 640" "256 320"}; do
  set -- $cfg
  echo "== WANN_POLLERS=$1 WANN_HANDOFF_COMPANION=$2"
  WANN_POLLERS=$1 WANN_HANDOFF_COMPANION=$2 timeout 900 python tools/frac_probe.py --fractions=${FRACS:--5,-6,-7,-8,-9,-11} --settings 80,1 --reps ${REPS:-4} 2>&1 | grep "^2\^" | sed -e 's/rounds.*handoffs/handoffs/' -e 's/packet_hops.*//' | cut -c1-200
done 2>&1 | tee $O/pollers.log
