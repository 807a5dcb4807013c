#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for tm in 4 2 4 2; do
  echo "=== bench bf16 1024 PW_TM=$tm"; LWG_BF16_PW_TM=$tm timeout 600 python bench.py --precision bf16 --size 1024 --workload novel_view --steps 3 --warmup 1 --no-extras --cpu-frames 0 --conv-breakdown 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('value', d['value'], 'ms/step', d['ms_per_step'], 'conv TF', r['achieved'], 'share', r['share_of_step_time'])"
  python - <<'PY'
import json
rows = json.load(open('gpurun_out/conv_breakdown.json'))
for r in rows:
    if 'taps1 ' in r['shape']: print('   ', r['shape'], round(1000 * r['ms'] / r['launches'], 1), 'us')
PY
done
echo "=== pw parity TM=2"; LWG_BF16_PW_TM=2 timeout 600 python - <<'PY' 2>&1 | tail -5
import sys
sys.path.insert(0, '.')
from tests import gpu_checks as g
r = g.check_bf16_conv_kernels()
for k, v in r.items():
    if k.startswith("1x1"): print(k, v)
print("OK")
PY
