#!/bin/bash
O=gpurun_out/methods
mkdir -p $O
C=/tmp/wann_methods_cache
for c in fenwick three_split; do
  python tools/bench_configs.py --config $c --threads '' --cache $C > $O/${c}_new.json 2> $O/${c}_new.log
  LD_LIBRARY_PATH=$PWD/tools/_scratch/base:$LD_LIBRARY_PATH python tools/bench_configs.py --config $c --threads '' --cache $C > $O/${c}_base.json 2> $O/${c}_base.log
done
