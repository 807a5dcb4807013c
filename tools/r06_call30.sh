#!/bin/bash
# Round-6 call 30: XCD-aware block order in the transposed Winograd kernel (tree; F(4x4,3x3) rule refined: N = 128 with Cin <= 128 keeps the chunked order) against variant -DLWG_CTW_XCD=0
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_an_convt_xcd_order.txt; : > $O
V=tools/lab/liblwg_ctw_xcd0.so
echo "== parity (tree)" >> $O
timeout 900 python -m pytest tests -q -m gpu -k "check_winograd4 or check_winograd_mode or check_whole_clip_batches or check_benched_shapes_512 or check_winograd_up4 or check_pipeline_full_512" 2>&1 | tail -3 >> $O
echo "== bench A/B (A = variant without the XCD order)" >> $O
tools/ab_bench.sh $V --steps 5 --warmup 2 >> $O 2>&1
cp ipercore_amd/liblwg_hip.so /tmp/liblwg_tree.so
for lib in tree xcd0; do
  if [ $lib = tree ]; then cp /tmp/liblwg_tree.so ipercore_amd/liblwg_hip.so; else cp $V ipercore_amd/liblwg_hip.so; fi
  timeout 600 python bench.py --no-extras --cpu-frames 0 --steps 4 --warmup 2 --conv-breakdown > /tmp/b.json 2>/dev/null
  cp gpurun_out/conv_breakdown.json gpurun_out/r06_an_breakdown_${lib}.json
done
cp /tmp/liblwg_tree.so ipercore_amd/liblwg_hip.so
python - >> $O <<'PY'
import json
a={r['shape']:r['ms']/r['launches'] for r in json.load(open('gpurun_out/r06_an_breakdown_tree.json'))}
b={r['shape']:r['ms']/r['launches'] for r in json.load(open('gpurun_out/r06_an_breakdown_xcd0.json'))}
print("== ms per launch inside the 300-frame step: XCD order | previous | ratio")
for k in a: print(f"{k:45s} {a[k]:8.3f} {b[k]:8.3f} {a[k]/b[k]:.3f}")
PY
cat $O
