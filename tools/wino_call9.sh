#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/w9; mkdir -p $O; L=tools/lab
for v in w1d2 w1d4 w2d1 w2d2 r3d1 r3d2; do echo "== $v"; timeout 100 python tools/winoshapes.py --lib $L/liblwg_w_$v.so --ts --only 1 2>&1 | grep "steady"; done > $O/cal.log 2>&1
cat $O/cal.log
