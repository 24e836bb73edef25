#!/bin/bash
# A/B: round-3 small core (tools/_scratch/base) vs the straight-line inner-product pass + counting merge + row request (working tree)
export TMPDIR=/tmp
O=gpurun_out/r04abwave
mkdir -p $O
: > $O/raw.log
for v in base new base new; do
  L=$LD_LIBRARY_PATH; [ $v = base ] && L=$PWD/tools/_scratch/base:$LD_LIBRARY_PATH
  for lp in 0 11904; do
    echo "== $v lean_pool=$lp" >> $O/raw.log
    LD_LIBRARY_PATH=$L WANN_LEAN_POOL=$lp python tools/phase_profile_small.py 1000000 40,80 20000 1:96 2>&1 | grep "wann raw" | awk '{print $4,$NF,$(NF-1)}' | tr '\n' ' ' >> $O/raw.log
    echo >> $O/raw.log
  done
  LD_LIBRARY_PATH=$L python tools/phase_profile_small.py 1000000 80 20000 0:128 2>&1 | grep "wann raw" | awk '{print "sift",$4,$NF,$(NF-1)}' | tr '\n' ' ' >> $O/raw.log
  echo >> $O/raw.log
done
: > $O/legs.log
for v in base new base new; do
  L=$LD_LIBRARY_PATH; [ $v = base ] && L=$PWD/tools/_scratch/base:$LD_LIBRARY_PATH
  for c in glove deep; do
    s=40,1; [ $c = deep ] && s=80,1
    echo "== $v $c" >> $O/legs.log
    LD_LIBRARY_PATH=$L WANN_DEEP_POLLERS=16 python tools/bench_configs.py --config $c --threads '' --cache /tmp/wann_cfg_cache --setting $s --seconds 4 2> $O/err.tmp | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config'], d['setting'], d['ms_per_batch'], d['search_kernel_ms'], d.get('k_search_tb_per_s'))" >> $O/legs.log
  done
done
