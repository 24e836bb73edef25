#!/bin/bash
O=gpurun_out/tiny
mkdir -p $O
: > $O/probe.log
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "brute or tiny or exact or leaf or golden or prefilter or dense or fenwick or three_split" > $O/tests.log 2>&1
for v in new base "new3"; do
echo "== $v" >> $O/probe.log
if [ $v = base ]; then export LD_LIBRARY_PATH=$PWD/tools/_scratch/base:$LD_LIBRARY_PATH; fi
if [ $v = new3 ]; then export LD_LIBRARY_PATH=$(echo $LD_LIBRARY_PATH | sed "s|$PWD/tools/_scratch/base:||"); export WANN_BRUTE_PER_CU=3; fi
python tools/frac_probe.py --fractions=-16,-15,-14,-13,-12 --settings 10,1 --reps 5 2>&1 | grep "^2\^" | cut -c1-80 >> $O/probe.log
WANN_PF_NO_REF=1 python tools/bench_prefilter.py 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('adverse scan_ms', d['scan_ms'], 'p12', d['synthetic_2pow_minus12']['device_ms'], 'mfma', d['device_ms'])" >> $O/probe.log
done
