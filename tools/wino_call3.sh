#!/bin/bash
# Round-5 lab call 3: knock-out timelines of the Winograd K loop (which class of filler work costs pipe time)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/w3; mkdir -p $O; L=tools/lab
for v in ko0 ko1 ko2 ko4 ko7 ko8; do echo "== $v"; for i in 1 3; do timeout 100 python tools/winoshapes.py --lib $L/liblwg_w_$v.so --ts --only $i 2>&1 | grep -v amdgpu.ids; done; done > $O/ko.log 2>&1
cat $O/ko.log
