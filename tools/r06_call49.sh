#!/bin/bash
# Round-6 call 49: bf16 halo-tile kernels: non-temporal output stores (variant -DLWG_BF16_NT_ST=1) against the tree, 1024 x 1024 bf16 novel view
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_bg_bf16_nt_stores.txt; : > $O
cp ipercore_amd/liblwg_hip.so /tmp/liblwg_tree.so
for rep in 1 2; do for v in variant tree; do
  if [ $v = tree ]; then cp /tmp/liblwg_tree.so ipercore_amd/liblwg_hip.so; else cp tools/lab/liblwg_bf16_ntst.so ipercore_amd/liblwg_hip.so; fi
  timeout 600 python bench.py --precision bf16 --size 1024 --workload novel_view --steps 4 --warmup 2 --no-extras --cpu-frames 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], 'fps  conv frac', d['roofline']['frac'], d.get('self_check'))" >> $O
done; done
cp /tmp/liblwg_tree.so ipercore_amd/liblwg_hip.so
cat $O
