#!/bin/bash
# Round-6 call 50: frame batch sweep of the 300-frame fp32 clip on the final tree
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_bh_frame_batch_sweep.txt; : > $O
for fb in 300 150 100 60 300; do
  timeout 600 python bench.py --no-extras --cpu-frames 0 --steps 4 --warmup 2 --frame-batch $fb 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('frame batch $fb:', d['value'], 'fps  conv frac', d['roofline']['frac'], d.get('self_check'))" >> $O
done
cat $O
