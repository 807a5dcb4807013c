#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/w7; mkdir -p $O; L=tools/lab
timeout 100 python tools/winoshapes.py --lib $L/liblwg_w_ts2.so --ts2 --only 1 2>&1 | grep -v amdgpu.ids > $O/ts2.log
cat $O/ts2.log
