#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r04wide4
mkdir -p $O
export WANN_PF_NO_REF=1 WANN_PF_DIM=512
rocprofv3 --kernel-trace --output-format csv -d $O/kt_new_trace -- python3 tools/bench_prefilter.py > $O/kt_new_trace.json 2> $O/kt_new_trace.log
python - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/r04wide4/kt_new_trace/runc/*kernel_trace.csv")[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "gemm_scores_wide" in r["Kernel_Name"] and int(r["End_Timestamp"])-int(r["Start_Timestamp"])>100000]
i=idx[-1]
t0=int(rows[i-9]["Start_Timestamp"])
with open("gpurun_out/r04wide4/last_call_timeline.txt","w") as o:
    for r in rows[i-9:i+8]:
        o.write(f'{(int(r["Start_Timestamp"])-t0)/1e3:9.1f} us  +{(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3:8.1f} us  {r["Kernel_Name"][:70]}\n')
PY
find $O -name '*kernel_trace.csv' -delete
