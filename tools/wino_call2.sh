#!/bin/bash
# Round-5 lab call 2: the re-scheduled Winograd kernel (tree library): correctness, per-shape times, phase timeline, bench in the mode
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
R=$PWD; O=gpurun_out/w4; mkdir -p $O; L=tools/lab
timeout 300 python tools/winolab.py ipercore_amd/liblwg_hip.so > $O/winolab.log 2>&1; echo "winolab exit=$?"
timeout 300 python tools/winoshapes.py > $O/shapes.log 2>&1; echo "shapes exit=$?"
for i in 0 1 3 5 8; do timeout 100 python tools/winoshapes.py --lib $L/liblwg_w_ts.so --ts --only $i; done > $O/ts.log 2>&1
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "winograd" 2>&1 | tail -5 > $O/pytest_wino.log
timeout 600 python bench.py --precision winograd --steps 5 --warmup 2 --cpu-frames 0 --no-extras > $O/bench_wino.json 2> $O/bench_wino.err; echo "bench exit=$?"
grep -v amdgpu.ids $O/winolab.log $O/shapes.log $O/ts.log; cat $O/pytest_wino.log; head -c 1500 $O/bench_wino.json
