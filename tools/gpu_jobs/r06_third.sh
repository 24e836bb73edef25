#!/bin/bash
# (WANN_INKERNEL_CAP: see r06_second.sh)
# round 6, third GPU call: what ends a mid-fraction batch NOW -- per-search timelines (TRACE build) at 2^-5 / 2^-6 / 2^-7 / 2^-9 with the
# default cap and with the four-wave kernel's cap lowered, phase cycles of the one-wave core alone and inside a 2^-9 batch (PROFILE build),
# the fixed multi-rank worker, FETCH_SIZE of every window fraction
export TMPDIR=/tmp
O=gpurun_out/r06c
mkdir -p $O
BASE_LD=$LD_LIBRARY_PATH
timeout 900 python -m pytest tests/test_distributed_gpu.py tests/test_production_mode.py -q -x -rs > $O/new_tests.log 2>&1
tail -5 $O/new_tests.log | cut -c1-300
export LD_LIBRARY_PATH=$PWD/tools/_scratch/trace:$BASE_LD
for cap in 0 320; do
  for p in -5 -6 -7 -9; do
    WANN_INKERNEL_CAP=$cap WANN_TASK_TRACE=$O/trace.txt timeout 600 python tools/frac_probe.py --fractions=$p --settings 80,1 --reps 1 > $O/probe${p}_cap$cap.log 2>&1
    python tools/trace_summary.py $O/trace.txt > $O/summary${p}_cap$cap.txt 2>&1
    rm -f $O/trace.txt
  done
done
export LD_LIBRARY_PATH=$PWD/tools/_scratch/prof:$BASE_LD
timeout 600 python tools/phase_profile.py 5120,2560,640 1 2>&1 | grep -v amdgpu.ids > $O/big_core_lone.log
WANN_PROFILE_PHASES=1 timeout 600 python tools/frac_probe.py --fractions=-9,-6 --settings 80,1 --reps 2 2>&1 | grep -v amdgpu.ids > $O/big_core_in_batch.log
export LD_LIBRARY_PATH=$BASE_LD
P="python3 bench.py --fractions headline --configs none --no-cpu-baseline --setting 80,1 --pipeline 0 --steps 5 --warmup 1"
for p in -16 -15 -14 -13 -12 -11 -10 -8 -7 -5 -4 -2 -1 0; do
  s=80,1; [ $p -ge -2 ] && s=40,1
  timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc${p}_g1 -- python3 bench.py --fractions headline --configs none --no-cpu-baseline --setting $s --pipeline 0 --steps 5 --warmup 1 --fraction $p > $O/pmc${p}_g1.json 2> $O/pmc${p}_g1.log
done
for f in $(find $O -name '*counter_collection.csv'); do head -1 $f > $f.sel.csv; grep "k_search\|k_brute" $f >> $f.sel.csv; rm -f $f; done
find $O -name '*agent_info.csv' -delete
for f in $O/summary*.txt; do echo "== $f"; head -14 $f | cut -c1-200; done
cat $O/big_core_lone.log | cut -c1-600 | tail -20
grep "companion phases\|^2\^" $O/big_core_in_batch.log | cut -c1-700
du -sh $O
