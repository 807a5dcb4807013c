#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/w6; mkdir -p $O; L=tools/lab
for v in kow0 kow1 kow2; do echo "== $v"; for i in 1 3; do timeout 100 python tools/winoshapes.py --lib $L/liblwg_w_$v.so --ts --only $i 2>&1 | grep -v amdgpu.ids; done; done > $O/kow.log 2>&1
cat $O/kow.log
