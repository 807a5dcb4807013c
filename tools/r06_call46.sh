#!/bin/bash
# Round-6 call 46: transposed Winograd kernel with the contiguous fragment panel [4][Cin/8][4][2][9 N] (tree) against the tree of the commit before it
# (tools/lab/wt_head: padded [N][12] panel): parity, determinism, whole bench A/B/A/B with per-shape times
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_bd_convt_panel.txt; : > $O
timeout 900 python -m pytest tests -q -m gpu -k "check_winograd_up4 or check_winograd_determinism or check_winograd_mode or check_whole_clip_batches or check_winograd_adversarial" 2>&1 | tail -2 >> $O
for rep in 1 2; do
  for v in old new; do
    if [ $v = old ]; then d=tools/lab/wt_head; else d=.; fi
    ( cd $d && timeout 600 python bench.py --no-extras --cpu-frames 0 --steps 5 --warmup 2 --conv-breakdown 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$v', d['value'], 'fps  up4', r.get('winograd_up4_kernel_frac'), d.get('self_check'))" ) >> $O
    cp $d/gpurun_out/conv_breakdown.json gpurun_out/r06_bd_breakdown_${v}_$rep.json
  done
done
python - >> $O <<'PY'
import json
def load(v): 
    acc={}
    for rep in (1,2):
        for r in json.load(open(f'gpurun_out/r06_bd_breakdown_{v}_{rep}.json')):
            acc.setdefault(r['shape'],[]).append(r['ms']/r['launches'])
    return {k:sum(x)/len(x) for k,x in acc.items()}
a,b=load('new'),load('old')
for k in a:
    if 'up2' in k: print(f"{k:45s} new {a[k]:8.3f} old {b[k]:8.3f} ratio {a[k]/b[k]:.3f}")
PY
cat $O
