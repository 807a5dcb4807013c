#!/bin/bash
# Round-6 call 38: F(4x4,3x3) epilogue layout B (a pass = half the channels of all 32 patches, every lane writes) against layout A (variant -DLWG_W4_EPIB=0):
# parity incl. the determinism check, A/B, per-shape times inside the step
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_au_wino4_epilogue_b.txt; : > $O
V=tools/lab/liblwg_w4_epia.so
echo "== parity (tree)" >> $O
timeout 900 python -m pytest tests -q -m gpu -k "check_winograd4 or check_winograd_mode or check_whole_clip_batches or check_winograd_determinism or check_winograd_adversarial or check_benched_shapes_512" 2>&1 | tail -3 >> $O
echo "== bench A/B (A = layout A variant)" >> $O
tools/ab_bench.sh $V --steps 5 --warmup 2 >> $O 2>&1
cat $O
