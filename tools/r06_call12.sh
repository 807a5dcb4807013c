#!/bin/bash
# Round-6 call 12: the F(4x4,3x3) kernel in the engine - kernel-level + adversarial + clip checks, then bench A/B (F(4x4,3x3) on / off) on one box
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 1200 python tools/gpu_diag.py check_winograd4 check_winograd_mode check_winograd_adversarial check_benched_shapes_512 check_whole_clip_batches check_generator_golden 2>&1 | grep -v amdgpu.ids | tail -12
cp $O/diag.json $O/r06_u_wino4_checks.json
for rep in 1 2; do
  for v in on off; do
    fl=""; [ $v = off ] && fl="--lab-no-wino4"
    timeout 600 python bench.py --no-extras --cpu-frames 0 --steps 8 --warmup 4 $fl 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('wino4 $v', d['value'], 'fps  conv frac', r['frac'], 'F(2,3)', r.get('winograd_kernel_frac'), 'F(4,3)', r.get('winograd4_kernel_frac'), 'up4', r.get('winograd_up4_kernel_frac'), 'alg-eq TF/s', r.get('algorithmic_equivalent_tflops'), d.get('self_check'))"
  done
done | tee $O/r06_u_ab_wino4_f32_512.txt
