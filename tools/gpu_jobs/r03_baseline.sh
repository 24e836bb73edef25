#!/bin/bash
# round-3 baseline evidence (before the kernel work): kernel stats of the mid window fractions, GloVe / deep legs with counters
export TMPDIR=/tmp
O=gpurun_out/r03base
mkdir -p $O
rocprofv3 -L > $O/counters.txt 2>&1
for p in -8 -11; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/frac$p -- python3 tools/frac_probe.py --fractions=$p --settings 80,1 --reps 5 > $O/frac$p.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/glove_kt -- python3 tools/bench_configs.py --config glove --threads '' > $O/glove_kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/glove_fetch -- python3 tools/bench_configs.py --config glove --threads '' > $O/glove_fetch.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/glove_sq -- python3 tools/bench_configs.py --config glove --threads '' > $O/glove_sq.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/deep_kt -- python3 tools/bench_configs.py --config deep --threads '' > $O/deep_kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/deep_fetch -- python3 tools/bench_configs.py --config deep --threads '' > $O/deep_fetch.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/deep_sq -- python3 tools/bench_configs.py --config deep --threads '' > $O/deep_sq.log 2>&1
# keep only the small summaries (kernel stats + counter csv of the search kernels)
find $O -name '*kernel_trace.csv' -size +2M -delete
for f in $(find $O -name '*counter_collection.csv'); do head -1 $f > $f.ksearch.csv; grep k_search $f >> $f.ksearch.csv; rm -f $f; done
du -sh $O
