#!/bin/bash
# round 6: k_finalize with one thread per output entry -- the GPU suite, then the headline and the GloVe-like leg (where the 14 - 21 us mattered most)
export TMPDIR=/tmp
O=gpurun_out/r06g
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1
tail -3 $O/gpu_tests.log | cut -c1-200
python bench.py --fractions headline --configs glove --no-cpu-baseline --setting 80,1 --steps 30 --warmup 5 > $O/bench.json 2> $O/bench.log
python3 - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r06g/bench.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms_per_step"], d["roofline"]["device_ms_per_step"], d["config"]["pipelined"])
g = d["configs"]["glove"]; print("glove", g["qps"], g["ms_per_batch"], g["search_kernel_ms"], g["roofline"]["frac"])
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $O/head_kt -- python3 bench.py --fractions headline --configs none --no-cpu-baseline --setting 80,1 --pipeline 0 --steps 30 --warmup 3 > $O/head_kt.json 2> $O/head_kt.log
grep "k_finalize\|k_route\|k_search" $O/head_kt/*/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-160
find $O -name '*kernel_trace.csv' -delete; find $O -name '*agent_info.csv' -delete
