#!/bin/bash
# round-3 evidence for profiles/: rocprofv3 kernel stats and PMC passes of the final code (separate passes for counters)
export TMPDIR=/tmp
O=gpurun_out/r03ev
mkdir -p $O
B="python3 bench.py --fractions headline --configs none --no-cpu-baseline --setting 80,1"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/head_kt -- $B --steps 50 --warmup 3 > $O/head_kt.json 2> $O/head_kt.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/head_fetch -- $B --steps 5 --warmup 1 > $O/head_fetch.json 2> $O/head_fetch.log
for p in -8 -11 -9; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/mid${p}_kt -- $B --fraction $p --steps 20 --warmup 2 > $O/mid${p}_kt.json 2> $O/mid${p}_kt.log
done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/mid-8_fetch -- $B --fraction -8 --steps 5 --warmup 1 > $O/mid-8_fetch.json 2> $O/mid-8_fetch.log
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $O/mid-8_sq -- $B --fraction -8 --steps 5 --warmup 1 > $O/mid-8_sq.json 2> $O/mid-8_sq.log
for c in glove deep; do
  s=40,1; [ $c = deep ] && s=80,1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${c}_kt -- python3 tools/bench_configs.py --config $c --threads '' --setting $s > $O/${c}_kt.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${c}_fetch -- python3 tools/bench_configs.py --config $c --threads '' --setting $s > $O/${c}_fetch.log 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/${c}_sq -- python3 tools/bench_configs.py --config $c --threads '' --setting $s > $O/${c}_sq.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prefilter_kt -- python3 tools/bench_prefilter.py > $O/prefilter_kt.json 2> $O/prefilter_kt.log
find $O -name '*kernel_trace.csv' -delete
for f in $(find $O -name '*counter_collection.csv'); do head -1 $f > $f.ksearch.csv; grep "k_search\|k_brute" $f >> $f.ksearch.csv; rm -f $f; done
du -sh $O
