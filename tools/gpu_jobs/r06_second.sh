#!/bin/bash
# (WANN_INKERNEL_CAP was a laboratory knob of this commit; the result is profiles/r06_inkernel_cap_sweep.txt and the knob is gone)
# round 6, second GPU call: (1) the tie-heavy parity test against the round-5 core (it must FAIL there), (2) the four-wave kernel's
# beam cap lowered (WANN_INKERNEL_CAP: levels above it run in the one-wave kernel with its helper waves), full counters per fraction,
# (3) the new multi-rank / production-mode tests, (4) the uint8 leg of the driver line
export TMPDIR=/tmp
O=gpurun_out/r06b
mkdir -p $O
BASE_LD=$LD_LIBRARY_PATH
LD_LIBRARY_PATH=$PWD/tools/_scratch/head:$BASE_LD timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "tie_heavy" > $O/tie_on_head.log 2>&1
tail -4 $O/tie_on_head.log | cut -c1-400
for cap in 0 640 320; do
  echo "== cap $cap" >> $O/caps.log
  WANN_INKERNEL_CAP=$cap timeout 900 python tools/frac_probe.py --fractions=-3,-4,-5,-6,-7,-8,-9,-10,-11 --settings 80,1 --reps 4 2>&1 | grep "^2\^" >> $O/caps.log
done
cat $O/caps.log | cut -c1-420
timeout 1500 python -m pytest tests/test_distributed_gpu.py tests/test_production_mode.py -q -x -rs > $O/new_tests.log 2>&1
tail -8 $O/new_tests.log
timeout 900 python tools/bench_configs.py --config sift_u8 --threads 32 --seconds 4 > $O/sift_u8.json 2> $O/sift_u8.log
tail -c 1500 $O/sift_u8.json
