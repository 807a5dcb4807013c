#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/w5; mkdir -p $O; L=tools/lab
timeout 300 python tools/winolab.py $L/liblwg_w_prio.so 2>&1 | grep -v amdgpu.ids > $O/winolab.log
timeout 300 python tools/winoshapes.py --lib $L/liblwg_w_prio.so --nodirect 2>&1 | grep -v amdgpu.ids > $O/shapes_prio.log
timeout 300 python tools/winoshapes.py --nodirect 2>&1 | grep -v amdgpu.ids > $O/shapes_tree.log
for i in 1 5; do timeout 100 python tools/winoshapes.py --lib $L/liblwg_w_priots.so --ts --only $i; done 2>&1 | grep -v amdgpu.ids > $O/ts_prio.log
cat $O/winolab.log $O/shapes_prio.log $O/shapes_tree.log $O/ts_prio.log
