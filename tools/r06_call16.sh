#!/bin/bash
# Round-6 call 16: F(4x4,3x3) kernel: weight / halo loads from a fixed (cache-hot) address - issue mechanics vs the memory system
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_z_wino4_fixed_addr.txt; : > $O
for v in tree UFIX HFIX UHFIX; do
  lib=""; [ $v != tree ] && lib="--lib tools/lab/liblwg_w4_$v.so"
  echo "== $v" >> $O
  for i in 0 3 5; do timeout 120 python tools/wino4lab.py $lib --w4only --only $i --frames 64 --reps 10 2>&1 | grep "F(4,3)" >> $O; done
done
grep -v worst $O
