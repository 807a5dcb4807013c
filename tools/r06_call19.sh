#!/bin/bash
# Round-6 call 19: the half-block form (4 waves, 4-channel stages, two workgroups per CU) of the F(4x4,3x3) kernel on EVERY plain / residual launch
# (variant library -DLWG_W4_SMALL=2) against the tree (half-block form for small launches only): per shape, parity, whole bench A/B/A/B
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_ac_wino4_half_block.txt; : > $O
V=tools/lab/liblwg_w4_half2.so
for lib in tree half2; do
  l=""; [ $lib != tree ] && l="--lib $V"
  echo "== $lib, 16 frames" >> $O
  timeout 300 python tools/wino4lab.py $l --w4only --frames 16 --reps 10 2>&1 | grep "F(4,3)\|sum" >> $O
  echo "== $lib, 64 frames" >> $O
  timeout 300 python tools/wino4lab.py $l --w4only --frames 64 --reps 5 2>&1 | grep "F(4,3)\|sum" >> $O
done
cp ipercore_amd/liblwg_hip.so /tmp/liblwg_tree.so
cp $V ipercore_amd/liblwg_hip.so
echo "== parity with the variant library" >> $O
timeout 900 python -m pytest tests -q -m gpu -k "check_winograd4 or check_winograd_mode or check_whole_clip_batches" 2>&1 | tail -5 >> $O
cp /tmp/liblwg_tree.so ipercore_amd/liblwg_hip.so
echo "== bench A/B (A = variant)" >> $O
tools/ab_bench.sh $V --steps 3 --warmup 1 >> $O 2>&1
cat $O
