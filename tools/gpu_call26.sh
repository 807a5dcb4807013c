#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for cfg in "--precision bf16" "--precision bf16 --size 256" "--precision split" "--precision bf16 --size 1024 --workload novel_view --streams 3 --no-conv-events" "--gather-dtype u8"; do
  echo "=== bench $cfg"; timeout 600 python bench.py $cfg --steps 3 --warmup 1 --no-extras --cpu-frames 0 2>gpurun_out/b.err | tail -1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); r = d.get('roofline') or {}
    print('value', d['value'], 'fb', d['config']['frame_batch'], 'frac', r.get('frac'), 'gov', r.get('frac_of_governing_roof'))
except Exception as e: print('FAILED', e)"; tail -2 gpurun_out/b.err | grep -v amdgpu.ids | cut -c1-300
done
