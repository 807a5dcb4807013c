#!/bin/bash
# PMC passes (each in its own rocprofv3 run, kernel-trace only - never combined with sys/hip/hsa traces):
#   pass A: MFMA busy / activity   pass B: FETCH_SIZE   pass C: WRITE_SIZE
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
R="$PWD"
CMD="python $R/bench.py --steps 3 --warmup 2 --cpu-frames 0 --no-conv-events --pipelined-streams 0 --output-frames 0 --no-split-extra"
run() { # name counters...
  local name=$1; shift
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$R/gpurun_out/pmc_$name" -o pmc -- $CMD > "$R/gpurun_out/pmc_$name.log" 2>&1 )
  echo "pmc $name exit=$?"; find "$R/gpurun_out/pmc_$name" -name "*.csv" | head -5
}
for p in ${PASSES:-A B C}; do
  case $p in
    A) run mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 ;;
    B) run fetch FETCH_SIZE ;;
    C) run write WRITE_SIZE ;;
    D) run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS ;;
  esac
done
# per-launch HBM-side traffic of the dominant kernel -> profiles/pmc_traffic.json (read by bench.py) + per-kernel tables
if [ -d gpurun_out/pmc_fetch ] && [ -d gpurun_out/pmc_write ]; then
  python tools/pmc_summary.py --traffic gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_traffic.json
fi
