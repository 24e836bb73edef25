#!/bin/bash
# round 6, first GPU call: (1) the raw beam-search parity tests incl. the new tie-heavy data, (2) same-box A/B of the tree against
# HEAD~ (tools/_scratch/head: the round-5 core without the exact tie handling), (3) VERDICT r05 item 1: batches in flight 1 / 2 / 4
# at 2^-9 and 2^-6, (4) counter passes of the 2^-9 and 2^-6 legs per kernel, (5) FETCH_SIZE on single-dword random probes
export TMPDIR=/tmp
O=gpurun_out/r06a
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -k "raw_beam_search or mid_fraction or final_research or lookahead" > $O/tests.log 2>&1
tail -3 $O/tests.log
BASE_LD=$LD_LIBRARY_PATH
for v in tree head tree head; do
  if [ $v = tree ]; then export LD_LIBRARY_PATH=$BASE_LD; else export LD_LIBRARY_PATH=$PWD/tools/_scratch/head:$BASE_LD; fi
  echo "== $v" >> $O/ab.log
  timeout 900 python tools/frac_probe.py --fractions=-3,-5,-6,-7,-8,-9,-10,-11 --settings 80,1 --reps 4 2>&1 | grep "^2\^" | cut -c1-120 >> $O/ab.log
done
export LD_LIBRARY_PATH=$BASE_LD
cat $O/ab.log
B="python3 bench.py --fraction -3 --configs none --no-cpu-baseline --setting 80,1 --steps 10 --warmup 2"
for n in 2 3 4; do
  timeout 900 $B --fractions=-9,-6 --pipeline $n > $O/inflight_$n.json 2> $O/inflight_$n.log
done
python3 - <<'PY'
import json
for n in (2, 3, 4):
    try:
        d = json.loads([l for l in open(f"gpurun_out/r06a/inflight_{n}.json") if l.startswith("{")][-1])
        print(n, "headline pipelined", d["config"]["pipelined"])
        for k, v in d["per_fraction"].items():
            print("  ", k, "blocking ms", v["device_ms"], "qps", v["qps"], "pipelined", v.get("pipelined"))
    except Exception as e:
        print(n, "failed", e)
PY
P="python3 bench.py --fractions headline --configs none --no-cpu-baseline --setting 80,1 --pipeline 0 --steps 5 --warmup 1"
for p in -9 -6; do
  i=0
  for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
             "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"; do
    i=$((i+1))
    timeout 600 rocprofv3 --pmc $grp --output-format csv -d $O/pmc${p}_g$i -- $P --fraction $p > $O/pmc${p}_g$i.json 2> $O/pmc${p}_g$i.log
  done
done
# single-dword probes from tables of 4 GiB (HBM) and 64 MiB (Infinity Cache)
for hot in 0 64; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/probe4_fetch_$hot -- tools/_bin/gather_calib 4 200000000 4 4 $hot > $O/probe4_fetch_$hot.json 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $O/probe4_tcc_$hot -- tools/_bin/gather_calib 4 200000000 4 4 $hot > $O/probe4_tcc_$hot.json 2>&1
done
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/r06a/*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        if "k_search" in kn or "k_probe4" in kn:
            acc[(kn.split("(")[0][-24:], r["Counter_Name"])].append(float(r["Counter_Value"]))
    print(f.split("/")[2])
    for k, v in sorted(acc.items()):
        v = sorted(v)
        print("   ", k, "median %.5g" % v[len(v) // 2], "n", len(v), "sum %.5g" % sum(v))
PY
for f in $(find $O -name '*counter_collection.csv'); do head -1 $f > $f.sel.csv; grep "k_search\|k_probe4" $f >> $f.sel.csv; rm -f $f; done
find $O -name '*agent_info.csv' -delete
du -sh $O
