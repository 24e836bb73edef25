#!/bin/bash
# round-4 measurement evidence: bench line (rotating batches), FETCH_SIZE calibration on k_search's access pattern, L2 hit rates
export TMPDIR=/tmp
O=gpurun_out/r04meas
mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "asynchronous or device_resident or default_stream" > $O/async_tests.log 2>&1
python bench.py --fractions headline --configs none --no-cpu-baseline --steps 20 --warmup 3 > $O/head.json 2> $O/head.log
rocprofv3 -L > $O/counters_list.txt 2>&1
G=tools/_bin/gather_calib
for rb in 512 448 384 256; do $G 4 34000000 $rb 10 >> $O/calib_plain.jsonl 2>> $O/calib.err; done
$G 4 34000000 512 10 64 >> $O/calib_plain.jsonl 2>> $O/calib.err
$G 4 34000000 512 10 3 >> $O/calib_plain.jsonl 2>> $O/calib.err
for rb in 512 384 256; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/calib_fetch_$rb -- $G 4 34000000 $rb 4 > $O/calib_fetch_$rb.json 2>> $O/calib.err
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $O/calib_tcc_$rb -- $G 4 34000000 $rb 4 > $O/calib_tcc_$rb.json 2>> $O/calib.err
done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/calib_fetch_hot64 -- $G 4 34000000 512 4 64 > $O/calib_fetch_hot64.json 2>> $O/calib.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/calib_fetch_hot3 -- $G 4 34000000 512 4 3 > $O/calib_fetch_hot3.json 2>> $O/calib.err
B="python3 bench.py --fractions headline --configs none --no-cpu-baseline --setting 80,1"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/head_fetch -- $B --steps 4 --warmup 1 > $O/head_fetch.json 2> $O/head_fetch.log
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $O/head_tcc -- $B --steps 4 --warmup 1 > $O/head_tcc.json 2> $O/head_tcc.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/head_kt -- $B --steps 50 --warmup 3 > $O/head_kt.json 2> $O/head_kt.log
for f in $(find $O -name '*counter_collection.csv'); do head -1 $f > $f.sel.csv; grep "k_search\|k_gather" $f >> $f.sel.csv; rm -f $f; done
find $O -name '*kernel_trace.csv' -size +2M -delete
du -sh $O
