#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/w11; mkdir -p $O; L=tools/lab
timeout 300 python tools/winolab.py ipercore_amd/liblwg_hip.so 2>&1 | grep -v amdgpu.ids > $O/winolab.log
timeout 300 python tools/winoshapes.py 2>&1 | grep -v amdgpu.ids > $O/shapes.log
for i in 1 5; do timeout 100 python tools/winoshapes.py --lib $L/liblwg_w_ts.so --ts --only $i; done 2>&1 | grep -v amdgpu.ids > $O/ts.log
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "winograd or generator_golden" 2>&1 | tail -3 > $O/pytest_wino.log
timeout 600 python bench.py --steps 5 --warmup 2 --cpu-frames 0 --no-extras > $O/bench.json 2> $O/bench.err; echo "bench exit=$?"
cat $O/winolab.log $O/shapes.log $O/ts.log $O/pytest_wino.log; head -c 400 $O/bench.json
bash tools/prof_pers.sh --use-vgg --use-face > $O/prof_pers_vgg_face.txt 2>&1; head -40 $O/prof_pers_vgg_face.txt
