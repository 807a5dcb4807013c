#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
R="$PWD"
for pass in A B; do
  if [ $pass = A ]; then C="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"; else C="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM"; fi
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$R/gpurun_out/pmc_bf16_hr2_$pass" -o pmc -- python "$R/tools/bf16lab.py" --no-f32 --convs-only --batch-mul 4 --shapes res64,skip0,up1,up2,gb128 --iters 4 > "$R/gpurun_out/pmc_bf16_hr2_$pass.log" 2>&1 )
  python tools/pmc_summary.py gpurun_out/pmc_bf16_hr2_$pass gpurun_out/pmc_bf16_hr2_$pass.md 2>/dev/null | grep -E "^\| kernel|lwg_conv_bf16" | cut -c1-420
done
