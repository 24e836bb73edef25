#!/bin/bash
# issue / instruction-cache / LDS counters of the small-beam core on a stand-alone inner-product graph (d = 96, beam 80, 10 000 queries)
export TMPDIR=/tmp
O=gpurun_out/r04spmc
mkdir -p $O
P="python3 tools/phase_profile_small.py 1000000 80 10000 1:96"
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" \
           "SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" \
           "SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_SALU SQ_IFETCH_LEVEL"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $O/g$i -- $P > $O/g$i.log 2>&1
done
for f in $(find $O -name '*counter_collection.csv'); do head -1 $f > $f.sel.csv; grep "k_search" $f >> $f.sel.csv; rm -f $f; done
du -sh $O
