#!/bin/bash
# Round-6 call 17 (and 25, with the SPADE epilogue in the small-launch form): the 4-wave small-launch form of the F(4x4,3x3) kernel: parity (cross-form bitwise) + frame batch 1 latency
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_ai_wino4_small_form_spade.txt; : > $O
timeout 900 python -m pytest tests -q -m gpu -k "check_winograd4 or check_winograd_mode or check_whole_clip_batches or check_winograd_adversarial" 2>&1 | tail -5 >> $O
timeout 600 python bench.py --steps 2 --warmup 1 --only-extras b1_latency --no-sizes-extra --cpu-frames 0 2>&1 | tail -1 > gpurun_out/r06_ai_bench_b1.json
python - <<'PY' >> $O
import json
l = json.loads(open("gpurun_out/r06_ai_bench_b1.json").read())
print("value", l.get("value"), "self_check", l.get("self_check"))
print(json.dumps(l.get("b1_latency"), indent=1))
PY
cat $O
