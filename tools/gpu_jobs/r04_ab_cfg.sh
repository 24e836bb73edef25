#!/bin/bash
# same-box A/B on the throughput-bound legs: headline (SIFT 2^-3), GloVe-like, deep-like -- tree build vs tools/_scratch/ab*
export TMPDIR=/tmp
O=gpurun_out/r04abcfg
mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "raw_beam_search or core_variants or index_matches or golden_reference" > $O/tests.log 2>&1
: > $O/ab.log
for v in tree $(ls -d tools/_scratch/ab* 2>/dev/null) tree; do
  if [ $v = tree ]; then L=$LD_LIBRARY_PATH; else L=$PWD/$v:$LD_LIBRARY_PATH; fi
  echo "== $v" >> $O/ab.log
  LD_LIBRARY_PATH=$L python bench.py --fractions headline --configs none --no-cpu-baseline --setting 80,1 --steps 20 --warmup 3 --pipeline 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sift', d['value'], d['roofline']['kernel_ms_per_step'], d['roofline']['frac'])" >> $O/ab.log
  for c in glove deep; do
    LD_LIBRARY_PATH=$L python tools/bench_configs.py --config $c --threads '' --cache /tmp/wann_cfg_cache 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config'], d['setting'], d['ms_per_batch'], d['search_kernel_ms'], d.get('k_search_tb_per_s'))" >> $O/ab.log
  done
done
