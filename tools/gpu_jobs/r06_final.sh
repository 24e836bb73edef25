#!/bin/bash
# round-6 evidence for profiles/: the GPU suite as the driver runs it (log kept), rocprofv3 kernel stats / PMC passes of the final tree
# (separate passes for counters), the configs[2..4] legs and the new driver-line legs under the profiler, the missing per-fraction
# FETCH_SIZE passes, a repetition stress of the mid-fraction batches
export TMPDIR=/tmp
O=gpurun_out/r06final
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q --durations=12 -rs > $O/gpu_tests.log 2>&1
tail -6 $O/gpu_tests.log | cut -c1-200
B="python3 bench.py --fractions headline --configs none --no-cpu-baseline --setting 80,1 --pipeline 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/head_kt -- $B --steps 50 --warmup 3 > $O/head_kt.json 2> $O/head_kt.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/head_fetch -- $B --steps 5 --warmup 1 > $O/head_fetch.json 2> $O/head_fetch.log
for p in -8 -11 -9 -6; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/mid${p}_kt -- $B --fraction $p --steps 20 --warmup 2 > $O/mid${p}_kt.json 2> $O/mid${p}_kt.log
done
# per-fraction FETCH_SIZE passes at the other settings the sweep picks on some boxes
for ps in "-8 40,1" "-8 160,1" "-10 160,1" "-11 160,1" "-2 80,1" "-7 160,1" "-9 160,1"; do
  set -- $ps
  timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc$1_$2_g1 -- python3 bench.py --fractions headline --configs none --no-cpu-baseline --setting $2 --pipeline 0 --steps 5 --warmup 1 --fraction $1 > $O/pmc$1_$2_g1.json 2> $O/pmc$1_$2_g1.log
done
export WANN_PF_NO_REF=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prefilter_kt -- python3 tools/bench_prefilter.py > $O/prefilter_kt.json 2> $O/prefilter_kt.log
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/prefilter_mfma -- python3 tools/bench_prefilter.py > $O/prefilter_mfma.json 2> $O/prefilter_mfma.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prefilter_fetch -- python3 tools/bench_prefilter.py > $O/prefilter_fetch.json 2> $O/prefilter_fetch.log
unset WANN_PF_NO_REF
python tools/bench_prefilter.py > $O/prefilter.json 2> $O/prefilter.log
for c in glove deep; do
  s=40,1; [ $c = deep ] && s=80,1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${c}_kt -- python3 tools/bench_configs.py --config $c --threads '' --cache /tmp/wann_cfg_cache --setting $s > $O/${c}_kt.json 2> $O/${c}_kt.log
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${c}_fetch -- python3 tools/bench_configs.py --config $c --threads '' --cache /tmp/wann_cfg_cache --setting $s > $O/${c}_fetch.json 2> $O/${c}_fetch.log
done
for cs in "fenwick 10,1" "three_split 20,1" "sift_u8 80,1"; do
  set -- $cs
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$1_kt -- python3 tools/bench_configs.py --config $1 --threads '' --cache /tmp/wann_cfg_cache --setting $2 > $O/$1_kt.json 2> $O/$1_kt.log
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/$1_fetch -- python3 tools/bench_configs.py --config $1 --threads '' --cache /tmp/wann_cfg_cache --setting $2 > $O/$1_fetch.json 2> $O/$1_fetch.log
done
for c in fenwick three_split; do
  python tools/bench_configs.py --config $c --threads 32 --cache /tmp/wann_cfg_cache > $O/$c.json 2> $O/$c.log
done
timeout 420 python tools/stress_repeat.py 300 > $O/stress_repeat.log 2>&1
tail -3 $O/stress_repeat.log | cut -c1-300
find $O -name '*kernel_trace.csv' -delete
for f in $(find $O -name '*counter_collection.csv'); do head -1 $f > $f.sel.csv; grep "k_search\|k_brute\|k_gemm\|k_rerank" $f >> $f.sel.csv; rm -f $f; done
find $O -name '*agent_info.csv' -delete
du -sh $O
