#!/bin/bash
# Round-6 call 47: transposed Winograd kernel: a k-pair's three weight loads spread over its MFMA slots (tree) against the burst (variant -DCTW_ULD_SPREAD=0)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_be_convt_uld_spread.txt; : > $O
timeout 600 python -m pytest tests -q -m gpu -k "check_winograd_up4 or check_winograd_determinism" 2>&1 | tail -2 >> $O
tools/ab_bench.sh tools/lab/liblwg_ctw_burst.so --steps 5 --warmup 2 >> $O 2>&1
cat $O
