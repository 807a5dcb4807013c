#!/bin/bash
# Round-6 call 13: knock-out builds of the F(4x4,3x3) kernel's K loop (what each filler class costs)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_v_wino4_knockouts.txt; : > $O
for v in tree KO_FRAG KO_ULD KO_HALO KO_TR KO_ALL; do
  lib=""; [ $v != tree ] && lib="--lib tools/lab/liblwg_w4_$v.so"
  echo "== $v" >> $O
  for i in 0 3 5; do timeout 120 python tools/wino4lab.py $lib --w4only --only $i --frames 64 --reps 10 2>&1 | grep "F(4,3)" >> $O; done
done
cat $O
