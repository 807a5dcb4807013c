#!/bin/bash
# Round-6 call 22: F(4x4,3x3) block order: chunks of gridDim.x tiles through every column block (tree) against column-block-major (variant -DLWG_W4_CHUNK=0)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_af_wino4_chunk_order.txt; : > $O
V=tools/lab/liblwg_w4_nochunk.so
for lib in tree nochunk; do
  l=""; [ $lib != tree ] && l="--lib $V"
  echo "== $lib, 128 frames" >> $O
  timeout 300 python tools/wino4lab.py $l --w4only --frames 128 --reps 5 2>&1 | grep "F(4,3)\|sum" >> $O
done
echo "== parity (tree)" >> $O
timeout 900 python -m pytest tests -q -m gpu -k "check_winograd4 or check_winograd_mode or check_whole_clip_batches" 2>&1 | tail -3 >> $O
echo "== bench A/B (A = column-block-major variant)" >> $O
tools/ab_bench.sh $V --steps 4 --warmup 2 >> $O 2>&1
cat $O
