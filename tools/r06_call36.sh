#!/bin/bash
# Round-6 call 36: transposed Winograd kernel after the store-hazard fix (pass offsets in the vector offset) + the exchange-slot permutation:
# determinism stress, parity, A/B against the library of commit 4828f76 (XCD order, linear slots, scalar pass offsets)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_as_convt_hazard_fix.txt; : > $O
echo "== determinism stress (tree)" >> $O
timeout 600 python tools/determinism_stress.py --reps 16 2>&1 | grep -v amdgpu.ids | tail -10 >> $O
echo "== parity (tree)" >> $O
timeout 900 python -m pytest tests -q -m gpu -k "check_winograd_up4 or check_winograd_mode or check_whole_clip_batches or check_winograd_adversarial or check_benched_shapes_512" 2>&1 | tail -3 >> $O
echo "== bench A/B (A = library of commit 4828f76)" >> $O
tools/ab_bench.sh tools/lab/liblwg_commit_xcd.so --steps 5 --warmup 2 >> $O 2>&1
cat $O
