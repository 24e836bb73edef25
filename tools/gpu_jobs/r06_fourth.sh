#!/bin/bash
# (WANN_SPEC_EXTRA was a laboratory knob of this commit: profiles/r06_extra_level_speculation.txt; the gated rule is a constant now, kSpecExtraLevels)
# round 6, fourth GPU call: one more speculated level for long chains (WANN_SPEC_EXTRA) per window fraction, its parity under the
# mid-fraction tests, the fenwick / three_split legs with the scans beside the searches, FETCH_SIZE of the new driver-line legs,
# bench.py under the launcher with one rank against the plain N = 1 run
export TMPDIR=/tmp
O=gpurun_out/r06d
mkdir -p $O
for v in 0 3 4 2 0 3; do
  echo "== spec_extra $v" >> $O/spec_extra.log
  WANN_SPEC_EXTRA=$v timeout 900 python tools/frac_probe.py --fractions=-2,-3,-4,-5,-6,-7,-8,-9,-10,-11 --settings 80,1 --reps 4 2>&1 | grep "^2\^" | cut -c1-200 >> $O/spec_extra.log
done
cat $O/spec_extra.log | cut -c1-140
WANN_SPEC_EXTRA=3 timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "mid_fraction or final_research or scheduling_variants or index_matches or lookahead or big_workgroup or golden_reference" > $O/tests_spec_extra.log 2>&1
tail -3 $O/tests_spec_extra.log
for c in fenwick three_split; do
  timeout 900 python tools/bench_configs.py --config $c --threads '' --cache /tmp/wann_cfg_cache > $O/$c.json 2> $O/$c.log
  python3 -c "
import json,sys
d=json.loads([l for l in open('$O/$c.json') if l.startswith('{')][-1])
print('$c', {k:d[k] for k in ('setting','ms_per_batch','qps','search_kernel_ms','algorithmic_gb_per_batch','k_search_tb_per_s','pipelined')})"
done
for c in fenwick three_split sift_u8; do
  s=$(python3 -c "
import json
d=json.loads([l for l in open('$O/$c.json') if l.startswith('{')][-1]) if '$c'!='sift_u8' else {'setting':{'beam':80,'mult':1}}
print('%d,%d'%(d['setting']['beam'],d['setting']['mult']))")
  timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${c}_fetch -- python3 tools/bench_configs.py --config $c --threads '' --cache /tmp/wann_cfg_cache --setting $s > $O/${c}_fetch.json 2> $O/${c}_fetch.log
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${c}_kt -- python3 tools/bench_configs.py --config $c --threads '' --cache /tmp/wann_cfg_cache --setting $s > $O/${c}_kt.json 2> $O/${c}_kt.log
done
B="bench.py --gpus 1 --steps 20 --warmup 5 --fractions headline --configs none --no-cpu-baseline --setting 80,1"
timeout 900 python $B > $O/plain_n1.json 2> $O/plain_n1.log
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 $B > $O/launcher_n1.json 2> $O/launcher_n1.log
python3 - <<'PY'
import json
a = json.loads([l for l in open("gpurun_out/r06d/plain_n1.json") if l.startswith("{")][-1])
b = json.loads([l for l in open("gpurun_out/r06d/launcher_n1.json") if l.startswith("{")][-1])
print("plain N=1", a["value"], a["ms_per_step"], "| under the launcher", b["value"], b["ms_per_step"], b["config"]["parallelism"][:40], "| ratio", round(b["value"] / a["value"], 4))
PY
find $O -name '*kernel_trace.csv' -delete
for f in $(find $O -name '*counter_collection.csv'); do head -1 $f > $f.sel.csv; grep "k_search\|k_brute" $f >> $f.sel.csv; rm -f $f; done
find $O -name '*agent_info.csv' -delete
for c in fenwick three_split sift_u8; do echo "== $c"; cat $O/${c}_kt/*/*kernel_stats.csv | cut -d, -f1-4,7-9 | head -8; done
du -sh $O
