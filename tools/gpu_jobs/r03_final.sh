#!/bin/bash
# the round's final evidence: the default bench line, then rocprofv3 kernel stats / the FETCH_SIZE pass of the same code
export TMPDIR=/tmp
O=gpurun_out/r03final
mkdir -p $O
python bench.py --steps 20 --warmup 3 > $O/bench_n1.json 2> $O/bench_n1.log
B="python3 bench.py --fractions headline --configs none --no-cpu-baseline --setting 80,1"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/head_kt -- $B --steps 50 --warmup 3 > $O/head_kt.json 2> $O/head_kt.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/head_fetch -- $B --steps 5 --warmup 1 > $O/head_fetch.json 2> $O/head_fetch.log
for p in -8 -11 -9; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/mid${p}_kt -- $B --fraction $p --steps 20 --warmup 2 > $O/mid${p}_kt.json 2> $O/mid${p}_kt.log
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prefilter_kt -- python3 tools/bench_prefilter.py > $O/prefilter_kt.json 2> $O/prefilter_kt.log
find $O -name '*kernel_trace.csv' -delete
for f in $(find $O -name '*counter_collection.csv'); do head -1 $f > $f.ksearch.csv; grep "k_search\|k_brute" $f >> $f.ksearch.csv; rm -f $f; done
du -sh $O
