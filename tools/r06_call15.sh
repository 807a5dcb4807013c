#!/bin/bash
# Round-6 call 15: F(4x4,3x3) kernel, weights in pairs + halo loads at k-pair 2 (tree) vs the first order (nt0 library)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_y4_wino4_paired_frags.txt; : > $O
timeout 200 python tools/wino4lab.py --parity 2>&1 | tail -2 >> $O
for rep in 1 2; do
for v in tree nt0; do
  lib=""; [ $v != tree ] && lib="--lib tools/lab/liblwg_w4_$v.so"
  echo "== $v" >> $O
  for i in 0 3 5 7; do timeout 120 python tools/wino4lab.py $lib --w4only --only $i --frames 64 --reps 10 2>&1 | grep "F(4,3)" >> $O; done
done
done
grep -v worst $O
