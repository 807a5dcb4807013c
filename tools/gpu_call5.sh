#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for bm in 1 4; do
  echo "=== bf16lab halo batch x$bm"; timeout 600 python tools/bf16lab.py --no-f32 --convs-only --batch-mul $bm 2>&1 | tee gpurun_out/bf16lab_v3_halo_bm$bm.txt | grep -v amdgpu.ids
done
echo "=== bench bf16 1024 fb=8 (halo)"; timeout 600 python bench.py --precision bf16 --size 1024 --workload novel_view --steps 3 --warmup 1 --no-extras --cpu-frames 0 --conv-breakdown 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('value', d['value'], 'ms/step', d['ms_per_step'], 'fb', d['config']['frame_batch'], 'conv TF', r['achieved'], 'share', r['share_of_step_time'], 'avg us', r['avg_launch_us'])"
cp gpurun_out/conv_breakdown.json gpurun_out/conv_breakdown_bf16_1024_halo.json
echo "=== bench bf16 1024 fb=8 HALO=0"; LWG_BF16_HALO=0 timeout 600 python bench.py --precision bf16 --size 1024 --workload novel_view --steps 3 --warmup 1 --no-extras --cpu-frames 0 2>&1 | tail -1 | cut -c1-160
echo "=== bf16 checks"; timeout 900 python - <<'PY' 2>&1 | tail -12
import sys, json, time
sys.path.insert(0, '.')
from tests import gpu_checks as g
for name in ("check_bf16_generator", "check_bf16_vs_oracle", "check_temporal_mode"):
    t0 = time.time()
    try:
        r = getattr(g, name)()
        print(name, "OK", round(time.time() - t0, 1), "s", json.dumps(r, default=str)[:1200], flush=True)
    except Exception as e:
        import traceback; traceback.print_exc()
        print(name, "FAILED", type(e).__name__, str(e)[:800], flush=True)
PY
for fb in 8 16 24; do
  echo "=== bench fp32 512 fb=$fb"; timeout 600 python bench.py --steps 3 --warmup 1 --no-extras --cpu-frames 0 --frame-batch $fb 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('value', d['value'], 'ms/step', d['ms_per_step'], 'conv TF', r['achieved'], 'frac', r['frac'], 'share', r['share_of_step_time'])"
done
