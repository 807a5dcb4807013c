#!/bin/bash
# Round-6 call 21: frame batch 1 with the half-block form (4-channel stages, 2 workgroups per CU) as the small-launch form
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
timeout 600 python bench.py --steps 2 --warmup 1 --only-extras b1_latency --no-sizes-extra --cpu-frames 0 2>&1 | tail -1 > gpurun_out/r06_ae_bench_b1.json
python - <<'PY'
import json
l = json.loads(open("gpurun_out/r06_ae_bench_b1.json").read())
print("value", l.get("value"), "self_check", l.get("self_check"))
b = l.get("b1_latency"); b.pop("roofline", None)
print(json.dumps(b, indent=1))
PY
