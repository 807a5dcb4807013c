#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
R="$PWD"
for pass in A B C; do
  case $pass in
    A) C="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" ;;
    B) C="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" ;;
    C) C="FETCH_SIZE TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum GRBM_GUI_ACTIVE" ;;
  esac
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$R/gpurun_out/pmc_attn_$pass" -o pmc -- python "$R/bench.py" --precision bf16 --size 1024 --workload novel_view --frames 32 --steps 1 --warmup 1 --no-extras --cpu-frames 0 --no-conv-events > "$R/gpurun_out/pmc_attn_$pass.log" 2>&1 )
  python tools/pmc_summary.py gpurun_out/pmc_attn_$pass gpurun_out/pmc_attn_$pass.md 2>/dev/null | grep -E "^\| kernel|attn|head_bf16|pw_kernel" | cut -c1-330
  find gpurun_out/pmc_attn_$pass -type f -size +3M -delete
done
