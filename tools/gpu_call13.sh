#!/bin/bash
# row-renaming bf16 kernel (lab A/B + parity), personalization with branch streams (time + trainer checks)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "=== bf16lab HR2 batch x4"; timeout 600 python tools/bf16lab.py --no-f32 --convs-only --batch-mul 4 2>&1 | tee gpurun_out/bf16lab_v8_hr2_bm4.txt | grep -v amdgpu.ids
echo "=== bf16lab HR (old) batch x4, 3x3 / 2x2 shapes"; LWG_BF16_HR2=0 timeout 600 python tools/bf16lab.py --no-f32 --convs-only --batch-mul 4 --shapes res64,skip0,up1,up2,gb128 2>&1 | grep -v amdgpu.ids
echo "=== bf16 checks"; timeout 900 python - <<'PY' 2>&1 | tail -6
import sys, json, time
sys.path.insert(0, '.')
from tests import gpu_checks as g
for name in ("check_bf16_generator", "check_bf16_vs_oracle"):
    t0 = time.time()
    try:
        r = getattr(g, name)()
        print(name, "OK", round(time.time() - t0, 1), "s", json.dumps(r, default=str)[:400], flush=True)
    except Exception as e:
        import traceback; traceback.print_exc()
        print(name, "FAILED", type(e).__name__, str(e)[:800], flush=True)
PY
echo "=== bench bf16 1024"; timeout 600 python bench.py --precision bf16 --size 1024 --workload novel_view --steps 3 --warmup 1 --no-extras --cpu-frames 0 --conv-breakdown 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('value', d['value'], 'ms/step', d['ms_per_step'], 'fb', d['config']['frame_batch'], 'conv TF', r['achieved'], 'share', r['share_of_step_time'], 'gov', r.get('frac_of_governing_roof'))"
cp gpurun_out/conv_breakdown.json gpurun_out/conv_breakdown_bf16_1024_v8.json
for v in "" "--no-branch-streams"; do
  echo "=== personalize $v"; timeout 600 python bench_personalize.py --steps 10 --warmup 3 $v 2>gpurun_out/pers.err | tail -1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); print('ms/step', d['ms_per_step'], 'TF', d['conv_tflops_whole_step'], 'host', d['single_step_host_enqueue_ms'], d['config']['step'][:30], 'loss', d['loss_G'], d['loss_D'])
except Exception as e: print('FAILED', e)"; grep -A8 "Raised at\|Fatal" gpurun_out/pers.err | head -12
done
echo "=== trainer tests"; timeout 1200 python -m pytest tests -m gpu -x -q -k "train or personal or discrim or vgg or face or backward or loss" 2>&1 | tail -4
