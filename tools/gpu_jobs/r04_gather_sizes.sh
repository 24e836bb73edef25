#!/bin/bash
# the gather ceiling against the size of the table gathered from (deep-like: 22.7 GiB of index; SIFT-1M: 3.3 GiB)
export TMPDIR=/tmp
O=gpurun_out/r04gsz
mkdir -p $O
G=tools/_bin/gather_calib
: > $O/sizes.jsonl
for gib in 1 4 8 16 24 48 96; do
  for rb in 512 448 384 256; do $G $gib 34000000 $rb 6 >> $O/sizes.jsonl 2>> $O/err.log; done
done
