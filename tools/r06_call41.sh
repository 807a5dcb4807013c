#!/bin/bash
# Round-6 call 41: F(4x4,3x3) cache policy of the output stores / halo loads alone (aux bits: 1 = sc0, 2 = nt, 16 = sc1) under the XCD-aware order
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_ay_wino4_cache_policy2.txt; : > $O
cp ipercore_amd/liblwg_hip.so /tmp/liblwg_tree.so
for v in tree w4_nt_st2 tree w4_nt_st2 w4_nt_st18 w4_nt_st3 tree w4_nt_st18; do
  if [ $v = tree ]; then cp /tmp/liblwg_tree.so ipercore_amd/liblwg_hip.so; else cp tools/lab/liblwg_$v.so ipercore_amd/liblwg_hip.so; fi
  timeout 300 python bench.py --no-extras --cpu-frames 0 --steps 4 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$v', d['value'], 'fps  w4', r.get('winograd4_kernel_frac'), d.get('self_check'))" >> $O
done
cp /tmp/liblwg_tree.so ipercore_amd/liblwg_hip.so
cat $O
