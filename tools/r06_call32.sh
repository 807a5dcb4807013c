#!/bin/bash
# Round-6 call 32: transposed Winograd kernel: pixel-slot permutation of the exchange buffer (conflict-free ds_write_b128 / ds_read_b128) against the linear order + an LDS counter pass
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_ap_convt_exchange_slots.txt; : > $O
echo "== parity (tree)" >> $O
timeout 900 python -m pytest tests -q -m gpu -k "check_winograd_up4 or check_winograd_mode or check_whole_clip_batches or check_winograd_adversarial" 2>&1 | tail -3 >> $O
echo "== bench A/B (A = linear slots variant)" >> $O
tools/ab_bench.sh tools/lab/liblwg_ctw_linear.so --steps 5 --warmup 2 >> $O 2>&1
rm -rf $O.pmc; mkdir -p gpurun_out/pmc_lds_rd
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d /root/repo/gpurun_out/pmc_lds_rd -o pmc -- python /root/repo/bench.py --steps 2 --warmup 1 --cpu-frames 0 --no-conv-events --no-extras --no-self-check > /root/repo/gpurun_out/pmc_lds_rd.log 2>&1 )
python tools/pmc_summary.py gpurun_out/pmc_lds_rd gpurun_out/pmc_lds_rd.md > /dev/null 2>&1
find gpurun_out/pmc_lds_rd -type f -size +3M -delete
head -8 gpurun_out/pmc_lds_rd.md | cut -c1-200 >> $O
cat $O
