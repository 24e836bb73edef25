#!/bin/bash
# round-5 evidence for profiles/: rocprofv3 kernel stats / PMC passes of the round's code (separate passes for counters), the
# configs[2..4] legs under the profiler, the mid core's probe and phase profile
export TMPDIR=/tmp
O=gpurun_out/r05final
mkdir -p $O
B="python3 bench.py --fractions headline --configs none --no-cpu-baseline --setting 80,1 --pipeline 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/head_kt -- $B --steps 50 --warmup 3 > $O/head_kt.json 2> $O/head_kt.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/head_fetch -- $B --steps 5 --warmup 1 > $O/head_fetch.json 2> $O/head_fetch.log
for p in -8 -11 -9 -6; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/mid${p}_kt -- $B --fraction $p --steps 20 --warmup 2 > $O/mid${p}_kt.json 2> $O/mid${p}_kt.log
done
export WANN_PF_NO_REF=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prefilter_kt -- python3 tools/bench_prefilter.py > $O/prefilter_kt.json 2> $O/prefilter_kt.log
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/prefilter_mfma -- python3 tools/bench_prefilter.py > $O/prefilter_mfma.json 2> $O/prefilter_mfma.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prefilter_fetch -- python3 tools/bench_prefilter.py > $O/prefilter_fetch.json 2> $O/prefilter_fetch.log
unset WANN_PF_NO_REF
python tools/bench_prefilter.py > $O/prefilter.json 2> $O/prefilter.log
WANN_PF_DIM=512 python tools/bench_prefilter.py > $O/prefilter_d512.json 2> $O/prefilter_d512.log
for c in glove deep; do
  s=40,1; [ $c = deep ] && s=80,1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${c}_kt -- python3 tools/bench_configs.py --config $c --threads '' --cache /tmp/wann_cfg_cache --setting $s > $O/${c}_kt.json 2> $O/${c}_kt.log
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${c}_fetch -- python3 tools/bench_configs.py --config $c --threads '' --cache /tmp/wann_cfg_cache --setting $s > $O/${c}_fetch.json 2> $O/${c}_fetch.log
done
for c in fenwick three_split; do
  python tools/bench_configs.py --config $c --threads 32 --cache /tmp/wann_cfg_cache > $O/$c.json 2> $O/$c.log
done
python tools/phase_profile.py 5120,2560 1 2>&1 | grep -v amdgpu.ids > $O/big_core_plain.log
find $O -name '*kernel_trace.csv' -delete
for f in $(find $O -name '*counter_collection.csv'); do head -1 $f > $f.sel.csv; grep "k_search\|k_brute\|k_gemm\|k_rerank" $f >> $f.sel.csv; rm -f $f; done
du -sh $O
