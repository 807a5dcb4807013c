#!/bin/bash
# Round-6 calls 27 / 28: F(4x4,3x3) LDS bank conflicts: epilogue reader lanes mapped to the ds_read_b128 lane groups (27), halo rows of 40 floats (28), each against its variant library + an LDS counter pass
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_al_wino4_rs40.txt; : > $O
echo "== parity (tree)" >> $O
timeout 900 python -m pytest tests -q -m gpu -k "check_winograd4 or check_winograd_mode or check_whole_clip_batches" 2>&1 | tail -3 >> $O
echo "== bench A/B (A = W4_RS 36 variant)" >> $O
tools/ab_bench.sh tools/lab/liblwg_w4_rs36.so --steps 5 --warmup 2 >> $O 2>&1
rm -rf $O.pmc; mkdir -p gpurun_out/pmc_lds_rd
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d /root/repo/gpurun_out/pmc_lds_rd -o pmc -- python /root/repo/bench.py --steps 2 --warmup 1 --cpu-frames 0 --no-conv-events --no-extras --no-self-check > /root/repo/gpurun_out/pmc_lds_rd.log 2>&1 )
python tools/pmc_summary.py gpurun_out/pmc_lds_rd gpurun_out/pmc_lds_rd.md > /dev/null 2>&1
find gpurun_out/pmc_lds_rd -type f -size +3M -delete
head -8 gpurun_out/pmc_lds_rd.md | cut -c1-200 >> $O
cat $O
