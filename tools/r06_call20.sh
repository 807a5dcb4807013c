#!/bin/bash
# Round-6 call 20: why the half-block form is slow: block timeline + knock-outs (halo, transform, weight loads; cache-hot halo)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_ad_wino4_half_block_why.txt; : > $O
for i in 1 3; do timeout 120 python tools/wino4lab.py --lib tools/lab/liblwg_w4_half2_ts.so --ts --half --only $i --frames 64 2>&1 | grep "\[ts\]\|epilogue:" >> $O; done
for v in half2 half2_kohalo half2_kotr half2_kould half2_hfix; do
  echo "== $v" >> $O
  for i in 0 1 7; do timeout 120 python tools/wino4lab.py --lib tools/lab/liblwg_w4_$v.so --w4only --only $i --frames 64 --reps 5 2>&1 | grep "F(4,3)" >> $O; done
done
cat $O
