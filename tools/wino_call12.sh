#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/w12; mkdir -p $O
timeout 300 python tools/winolab.py ipercore_amd/liblwg_hip.so 2>&1 | grep -v amdgpu.ids > $O/winolab.log
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "winograd or generator_golden or check_benched_shapes_512 or check_pipeline_tiny_64" 2>&1 | tail -5 > $O/pytest_wino.log
for f in 1 2 4 16; do timeout 200 python tools/winoshapes.py --frames $f --nodirect 2>&1 | grep -v amdgpu.ids | tail -1; done > $O/frames.log
timeout 600 python bench.py --steps 5 --warmup 2 --cpu-frames 0 --no-split-extra --no-sizes-extra --output-frames 0 > $O/bench.json 2> $O/bench.err; echo "bench exit=$?"
cat $O/winolab.log $O/pytest_wino.log $O/frames.log; python - <<'PY'
import json
d=json.loads(open('gpurun_out/w12/bench.json').read().split("\n")[0])
print(d["value"], d["self_check"], d["roofline"]["frac"], "b1", d.get("b1_latency",{}).get("ms_per_frame_back_to_back"), d.get("b1_latency",{}).get("roofline",{}).get("frac"), "shard", d.get("shard_of_8",{}).get("t_shard_ms"))
PY
