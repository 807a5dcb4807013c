#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/w10; mkdir -p $O; L=tools/lab
timeout 300 python tools/winolab.py ipercore_amd/liblwg_hip.so 2>&1 | grep -v amdgpu.ids > $O/winolab.log
timeout 300 python tools/winoshapes.py 2>&1 | grep -v amdgpu.ids > $O/shapes.log
for i in 1 3 5; do timeout 100 python tools/winoshapes.py --lib $L/liblwg_w_ts.so --ts --only $i; done 2>&1 | grep -v amdgpu.ids > $O/ts.log
timeout 100 python tools/winoshapes.py --lib $L/liblwg_w_ts2.so --ts2 --only 1 2>&1 | grep -v amdgpu.ids > $O/ts2.log
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "winograd" 2>&1 | tail -3 > $O/pytest_wino.log
timeout 600 python bench.py --precision winograd --steps 5 --warmup 2 --cpu-frames 0 --no-extras > $O/bench_wino.json 2> $O/bench_wino.err; echo "bench exit=$?"
cat $O/winolab.log $O/shapes.log $O/ts.log; head -14 $O/ts2.log; tail -3 $O/ts2.log; cat $O/pytest_wino.log; head -c 600 $O/bench_wino.json
