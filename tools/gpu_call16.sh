#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
R="$PWD"
for v in "" "--no-overlap-d"; do
  echo "=== personalize $v"; timeout 600 python bench_personalize.py --steps 10 --warmup 3 $v 2>gpurun_out/pers.err | tail -1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); print('ms/step', d['ms_per_step'], 'TF', d['conv_tflops_whole_step'], 'host', d['single_step_host_enqueue_ms'], d['config']['step'][:60], 'loss', d['loss_G'], d['loss_D'])
except Exception as e: print('FAILED', e)"; grep -A8 "Raised at\|Fatal" gpurun_out/pers.err | head -12
done
echo "=== trainer tests"; timeout 1200 python -m pytest tests -m gpu -x -q -k "train or personal or discrim or vgg or face or backward or loss" 2>&1 | tail -4
