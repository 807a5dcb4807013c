#!/bin/bash
# usage: ab_bench.sh LIB_A.so [bench args...]: bench.py with library A and with the tree's library, A/B/A/B in one call on one box (boxes differ by +-5 %)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
A=$1; shift
cp ipercore_amd/liblwg_hip.so /tmp/liblwg_tree.so
for rep in 1 2; do
  for lib in A tree; do
    if [ $lib = tree ]; then cp /tmp/liblwg_tree.so ipercore_amd/liblwg_hip.so; else cp $A ipercore_amd/liblwg_hip.so; fi
    timeout 600 python bench.py --no-extras --cpu-frames 0 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$lib', d['value'], 'fps  conv frac', r['frac'], 'wino', r.get('winograd_kernel_frac'), 'up4', r.get('winograd_up4_kernel_frac'), d.get('self_check'))"
  done
done
cp /tmp/liblwg_tree.so ipercore_amd/liblwg_hip.so
