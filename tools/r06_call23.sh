#!/bin/bash
# Round-6 call 23: per-shape conv time inside the bench (300-frame clip) for both block orders
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
V=tools/lab/liblwg_w4_nochunk.so
cp ipercore_amd/liblwg_hip.so /tmp/liblwg_tree.so
for lib in tree nochunk tree nochunk; do
  if [ $lib = tree ]; then cp /tmp/liblwg_tree.so ipercore_amd/liblwg_hip.so; else cp $V ipercore_amd/liblwg_hip.so; fi
  timeout 600 python bench.py --no-extras --cpu-frames 0 --steps 4 --warmup 2 --conv-breakdown > /tmp/b.json 2>/dev/null
  python -c "
import json
d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1]); print('$lib', d['value'])"
  cp gpurun_out/conv_breakdown.json gpurun_out/r06_ag_breakdown_${lib}_$RANDOM.json
done
cp /tmp/liblwg_tree.so ipercore_amd/liblwg_hip.so
