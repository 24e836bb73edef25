#!/bin/bash
# round 6: the GPU suite as the driver runs it, smoke, then the default bench line (all fractions, all configs, cpu baseline)
export TMPDIR=/tmp
O=gpurun_out/r06val
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q --durations=12 -rs ${PYTEST_SELECT:-} > $O/gpu_tests.log 2>&1
tail -22 $O/gpu_tests.log | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
SECONDS=0; python bench.py --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.log; echo "bench.py wall: $SECONDS s"

python3 - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r06val/bench_n1.json") if l.startswith("{")][-1])
print({k: d[k] for k in ("metric", "value", "ms_per_step")}, d["roofline"]["frac"], d["roofline"]["traffic"])
print("cpu", d.get("cpu_baseline"))
for k, v in d["per_fraction"].items():
    print(k, v["beam"], v["mult"], v["device_ms"], v["roofline_frac"], v.get("traffic_over_algorithmic"), v.get("rows_identical_ids"), v.get("pipelined", {}).get("speedup_over_blocking"))
for k, v in d["configs"].items():
    print(k, {a: v.get(a) for a in ("qps", "ms_per_batch", "search_kernel_ms", "leg_wall_s", "error")}, v.get("roofline"), [(r.get("threads"), round(r.get("qps", 0)), r.get("same_ids")) for r in v.get("reference", [])] if isinstance(v.get("reference"), list) else v.get("reference"))
PY
