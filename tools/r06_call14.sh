#!/bin/bash
# Round-6 call 14: F(4x4,3x3) kernel with the contiguous unpadded panel; non-temporal activation traffic on / off
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_w_wino4_panel_nt.txt; : > $O
timeout 200 python tools/wino4lab.py --parity 2>&1 | tail -3 >> $O
for v in tree nt0; do
  lib=""; [ $v != tree ] && lib="--lib tools/lab/liblwg_w4_$v.so"
  echo "== $v" >> $O
  for i in 0 3 5 7; do timeout 120 python tools/wino4lab.py $lib --w4only --only $i --frames 64 --reps 10 2>&1 | grep "F(4,3)" >> $O; done
done
grep -v worst $O
