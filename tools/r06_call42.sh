#!/bin/bash
# Round-6 call 42: non-temporal output stores in both Winograd kernels (tree) against the transposed kernel with default stores (variant -DCTW_NT_ST=0): determinism, parity, A/B
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_az_nt_stores.txt; : > $O
timeout 600 python tools/determinism_stress.py --reps 8 2>&1 | grep -v amdgpu.ids | tail -1 >> $O
timeout 900 python -m pytest tests -q -m gpu -k "check_winograd4 or check_winograd_up4 or check_winograd_mode or check_whole_clip_batches or check_benched_shapes_512" 2>&1 | tail -2 >> $O
echo "== bench A/B (A = transposed kernel with default stores)" >> $O
tools/ab_bench.sh tools/lab/liblwg_ctw_st0.so --steps 5 --warmup 2 >> $O 2>&1
cat $O
