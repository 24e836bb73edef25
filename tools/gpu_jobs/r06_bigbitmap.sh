#!/bin/bash
# round 6: the one-wave core's bitmap of written filter-slot groups (probes / touches of unwritten groups skipped): parity tests,
# same-box A/B against the commit before (tools/_scratch/head), FETCH_SIZE of the 2^-9 leg
export TMPDIR=/tmp
O=gpurun_out/r06e
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -x -k "raw_beam_search or mid_fraction or final_research or lookahead or scheduling_variants or big_workgroup or unserved or sift_1m or deep_chains" > $O/tests.log 2>&1
tail -3 $O/tests.log
BASE_LD=$LD_LIBRARY_PATH
for v in tree head tree head; do
  if [ $v = tree ]; then export LD_LIBRARY_PATH=$BASE_LD; else export LD_LIBRARY_PATH=$PWD/tools/_scratch/head:$BASE_LD; fi
  echo "== $v" >> $O/ab.log
  timeout 900 python tools/frac_probe.py --fractions=-3,-6,-7,-8,-9,-10,-11 --settings 80,1 --reps 4 2>&1 | grep "^2\^" | cut -c1-130 >> $O/ab.log
done
export LD_LIBRARY_PATH=$BASE_LD
cat $O/ab.log
timeout 600 python tools/phase_profile.py 5120,2560 1 2>&1 | grep "wann raw\|nq=" > $O/lone_tree.log
LD_LIBRARY_PATH=$PWD/tools/_scratch/head:$BASE_LD timeout 600 python tools/phase_profile.py 5120,2560 1 2>&1 | grep "wann raw\|nq=" > $O/lone_head.log
echo "lone searches, tree:"; cat $O/lone_tree.log; echo "head:"; cat $O/lone_head.log
for p in -9 -8; do
  timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc${p}_g1 -- python3 bench.py --fractions headline --configs none --no-cpu-baseline --setting 80,1 --pipeline 0 --steps 5 --warmup 1 --fraction $p > $O/pmc${p}_g1.json 2> $O/pmc${p}_g1.log
done
for f in $(find $O -name '*counter_collection.csv'); do head -1 $f > $f.sel.csv; grep "k_search\|k_brute" $f >> $f.sel.csv; rm -f $f; done
find $O -name '*agent_info.csv' -delete
python3 tools/summarize_mid_pmc.py $O r06x -9 -8 && rm -f profiles/r06x_*
