#!/bin/bash
# lab: MFMA-busy / activity counters of the personalization step (one rocprofv3 --pmc pass with the kernel trace only)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
rm -rf gpurun_out/pmc_pers
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/pmc_pers -o pmc -- python $R/bench_personalize.py --steps 3 --warmup 1 > $R/gpurun_out/pmc_pers.log 2>&1 ); echo "pmc pers exit=$?"
python tools/pmc_summary.py gpurun_out/pmc_pers gpurun_out/pmc_pers.md > /dev/null 2>&1
find gpurun_out/pmc_pers -type f -size +3M -delete
head -12 gpurun_out/pmc_pers.md
