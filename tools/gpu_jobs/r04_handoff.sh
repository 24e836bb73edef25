#!/bin/bash
# mid window fractions: at most half the pollers run look-aheads asked for on evidence; chains beyond their levels go to idle pollers from 8 x the first beam
# (WANN_LA_SPEC_CAP / WANN_COMPANION_HANDOFF belong to an experimental build that was reverted -- DESIGN.md section 7, open item 1; the
#  shipped library ignores them.  Kept as the record of profiles/r04_mid_fraction_schedule_sweeps.txt.)
export TMPDIR=/tmp
O=gpurun_out/r04handoff
mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mid_fraction or lookahead or deep_chains or unserved or final_research or serialised" > $O/tests2.log 2>&1
: > $O/cap.log
run() {
  echo "== $*" >> $O/cap.log
  env "$@" python tools/frac_probe.py --fractions=-5,-6,-7,-8,-9,-10,-11 --settings 80,1 --reps 3 2>&1 | grep "^2\^" >> $O/cap.log
}
run A=1
run WANN_LA_SPEC_CAP=1000 WANN_COMPANION_HANDOFF=0
run WANN_LA_SPEC_CAP=8
run WANN_LA_SPEC_CAP=24
run A=2
run WANN_LA_SPEC_CAP=1000 WANN_COMPANION_HANDOFF=0
run WANN_SCAN_NUM=8
run WANN_SCAN_NUM=12
