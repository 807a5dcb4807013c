#!/bin/bash
# Round-6 call 9: the fused bf16 up4 + head launch with the lighter epilogue (parity, lab time, A/B in the step); default bench with the B = 1 stream numbers
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 900 python tools/gpu_diag.py check_bf16_up4_head check_bf16_generator check_benched_shapes_1024_bf16 2>&1 | grep -v amdgpu.ids | tail -5
timeout 300 python tools/up4headlab.py 2>&1 | grep -v amdgpu.ids | tee $O/r06_g_up4headlab.txt
timeout 900 python tools/bf16_ab.py 4 2>&1 | grep -v amdgpu.ids | tee $O/r06_g_bf16_up4_head_ab.txt
timeout 900 python bench.py --no-sizes-extra --no-split-extra --output-frames 0 2>/dev/null | tail -1 > $O/r06_g_bench.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06_g_bench.json"))
print(d["value"], "fps")
print(json.dumps(d["b1_latency"])[:900])
PY
