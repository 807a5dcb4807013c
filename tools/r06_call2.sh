#!/bin/bash
# Round-6 call 2: the persistent Winograd kernels (parity, then A/B against the one-block-per-workgroup build, A/B/A/B in one call on one box),
# the extended bisect of the background network's gradient outliers.
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 1500 python tools/gpu_diag.py check_winograd_mode check_winograd_up4 check_generator_golden check_pipeline_full_512 check_benched_shapes_512 check_winograd_adversarial 2>&1 | grep -v amdgpu.ids | tail -9
cp $O/diag.json $O/r06_b_persistent_checks.json 2>/dev/null
cp ipercore_amd/liblwg_hip.so /tmp/liblwg_head.so
for rep in 1 2; do
  for lib in nopersist head; do
    if [ $lib = head ]; then cp /tmp/liblwg_head.so ipercore_amd/liblwg_hip.so; else cp tools/lab/liblwg_nopersist.so ipercore_amd/liblwg_hip.so; fi
    timeout 600 python bench.py --steps 8 --warmup 4 --no-extras --cpu-frames 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$lib', d['value'], 'fps  conv frac', r['frac'], 'wino', r.get('winograd_kernel_frac'), 'up4', r.get('winograd_up4_kernel_frac'), d.get('self_check'))"
  done
done
cp /tmp/liblwg_head.so ipercore_amd/liblwg_hip.so
timeout 1500 python tools/diag_bg_grads.py 512 3 2>&1 | grep -v amdgpu.ids > $O/r06_b_bg_grads_bisect.txt; tail -75 $O/r06_b_bg_grads_bisect.txt | cut -c1-330
