#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/deepprobe
mkdir -p $O
python tools/bench_configs.py --config deep --threads '' --setting 80,1 > $O/plain.log 2>&1
rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum --output-format csv -d $O/tlb -- python3 tools/bench_configs.py --config deep --threads '' --setting 80,1 > $O/tlb.log 2>&1
rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum --output-format csv -d $O/tlb_sift -- python3 bench.py --steps 5 --warmup 1 --fractions headline --configs none --no-cpu-baseline --setting 80,1 > $O/tlb_sift.log 2>&1
for f in $(find $O -name '*counter_collection.csv'); do head -1 $f > $f.ksearch.csv; grep k_search $f >> $f.ksearch.csv; rm -f $f; done
LD_LIBRARY_PATH=$PWD/tools/_scratch/trace:$LD_LIBRARY_PATH WANN_TASK_TRACE=$O/trace_deep.txt python tools/bench_configs.py --config deep --threads '' --setting 80,1 > $O/trace.log 2>&1
python tools/trace_summary.py $O/trace_deep.txt > $O/trace_summary.txt 2>&1
rm -f $O/trace_deep.txt
