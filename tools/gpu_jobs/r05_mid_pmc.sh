#!/bin/bash
# round 5: instruction mix per hop of the four-wave kernel's general core (third-generation in the tree, first-generation in
# tools/_scratch/ab1) on a stand-alone L2 graph, 8192 searches at beam 640
export TMPDIR=/tmp
O=gpurun_out/r05pmc
mkdir -p $O
P="python3 tools/mid_core_probe.py 1000000 ${BEAM:-640} 8192"
BASE_LD=$LD_LIBRARY_PATH
for v in tree ab1; do
  if [ $v = tree ]; then export LD_LIBRARY_PATH=$BASE_LD; else export LD_LIBRARY_PATH=$PWD/tools/_scratch/ab1:$BASE_LD; fi
  i=0
  for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" \
             "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU"; do
    i=$((i+1))
    rocprofv3 --pmc $grp --output-format csv -d $O/${v}_g$i -- $P > $O/${v}_g$i.log 2>&1
  done
done
export LD_LIBRARY_PATH=$BASE_LD
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/r05pmc/*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_search" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f.split("/")[2], {k: "%.4g" % (sum(v) / len(v)) for k, v in acc.items()}, "dispatch rows", {k: len(v) for k, v in acc.items()})
PY
find $O -name '*counter_collection.csv' -size +2M -delete
