#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "=== bf16 kernel checks"; timeout 900 python - <<'PY' 2>&1 | tail -30
import sys, json, time
sys.path.insert(0, '.')
from tests import gpu_checks as g
for name in ("check_bf16_conv_kernels", "check_bf16_generator"):
    t0 = time.time()
    try:
        r = getattr(g, name)()
        print(name, "OK", round(time.time() - t0, 1), "s", flush=True)
        for k, v in r.items(): print("   ", k, v)
    except Exception as e:
        import traceback; traceback.print_exc()
        print(name, "FAILED", type(e).__name__, str(e)[:800], flush=True)
PY
echo "=== bf16lab up-sampling, fused"; timeout 300 python tools/bf16lab.py --no-f32 --convs-only --batch-mul 4 --shapes up2,up1,up0 2>&1 | grep -v amdgpu.ids
echo "=== bf16lab up-sampling, four launches"; LWG_LAB_UP4=0 timeout 300 python tools/bf16lab.py --no-f32 --convs-only --batch-mul 4 --shapes up2,up1,up0 2>&1 | grep -v amdgpu.ids
echo "=== bench bf16 1024"; timeout 600 python bench.py --precision bf16 --size 1024 --workload novel_view --steps 3 --warmup 1 --no-extras --cpu-frames 0 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('value', d['value'], 'ms/step', d['ms_per_step'], 'fb', d['config']['frame_batch'], 'conv TF', r['achieved'], 'share', r['share_of_step_time'], 'gov', r.get('frac_of_governing_roof'))"
echo "=== bf16 vs oracle"; timeout 600 python - <<'PY' 2>&1 | tail -3
import sys, json
sys.path.insert(0, '.')
from tests import gpu_checks as g
print(json.dumps(g.check_bf16_vs_oracle(), default=str)[:600])
PY
