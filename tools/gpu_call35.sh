#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for fb in 8 12 16; do
  echo "=== bf16 1024 fb=$fb"; timeout 600 python bench.py --precision bf16 --size 1024 --workload novel_view --steps 3 --warmup 1 --no-extras --cpu-frames 0 --frame-batch $fb 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('value', d['value'], 'ms/step', d['ms_per_step'], 'conv TF', r['achieved'], 'share', r['share_of_step_time'])"
done
for fb in 16 24 32; do
  echo "=== fp32 512 fb=$fb"; timeout 600 python bench.py --steps 4 --warmup 2 --no-extras --cpu-frames 0 --frame-batch $fb 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('value', d['value'], 'ms/step', d['ms_per_step'], 'conv TF', r['achieved'], 'share', r['share_of_step_time'])"
done
