#!/bin/bash
# round 5, final code: the GPU suite as the driver runs it, then the default bench line
export TMPDIR=/tmp
O=gpurun_out/r05val
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1
tail -4 $O/gpu_tests.log | head -3
python bench.py --steps 20 --warmup 3 > $O/bench_n1.json 2> $O/bench_n1.log
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r05val/bench_n1.json"))
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"], d["roofline"]["kernel_ms_per_step"], d["config"]["pipelined"])
for k,v in d["per_fraction"].items(): print(k, v["qps"], v["beam"], v["mult"], v["device_ms"], v["roofline_frac"], v.get("rows_identical_dists"), v.get("rows_identical_ids"), v.get("pipelined"))
for k,v in d["configs"].items(): print(k, v.get("qps"), (v.get("roofline") or {}).get("frac"), (v.get("roofline") or {}).get("traffic"), v.get("pipelined"))
print(d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["host_cores"], d["cpu_baseline"]["gpu_rows_identical_ids"])
PY
