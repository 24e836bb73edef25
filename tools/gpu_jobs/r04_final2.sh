#!/bin/bash
# round-4 evidence after the lean-pool fix: the default bench line, rocprofv3 kernel stats of the headline and of the inner-product legs
export TMPDIR=/tmp
O=gpurun_out/r04final2
mkdir -p $O
python bench.py --steps 20 --warmup 3 > $O/bench_n1.json 2> $O/bench_n1.log
B="python3 bench.py --fractions headline --configs none --no-cpu-baseline --setting 80,1 --pipeline 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/head_kt -- $B --steps 50 --warmup 3 > $O/head_kt.json 2> $O/head_kt.log
for c in glove deep; do
  s=40,1; [ $c = deep ] && s=80,1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${c}_kt -- python3 tools/bench_configs.py --config $c --threads '' --cache /tmp/wann_cfg_cache --setting $s > $O/${c}_kt.json 2> $O/${c}_kt.log
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${c}_fetch -- python3 tools/bench_configs.py --config $c --threads '' --cache /tmp/wann_cfg_cache --setting $s --seconds 2 > $O/${c}_fetch.json 2> $O/${c}_fetch.log
done
find $O -name '*kernel_trace.csv' -delete
for f in $(find $O -name '*counter_collection.csv'); do head -1 $f > $f.sel.csv; grep "k_search" $f >> $f.sel.csv; rm -f $f; done
du -sh $O
