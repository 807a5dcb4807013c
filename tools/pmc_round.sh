#!/bin/bash
# PMC passes (each in its own rocprofv3 run, kernel-trace only - never combined with sys/hip/hsa traces):
#   pass A: MFMA busy / activity   pass B: FETCH_SIZE   pass C: WRITE_SIZE
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
R="$PWD"
# BENCH_ARGS: extra bench.py flags (e.g. "--precision bf16 --size 1024 --workload novel_view"); TAG: suffix of the output names;
# KERNEL: name substring of the kernel family whose per-launch traffic goes to gpurun_out/pmc_traffic$TAG.json
CMD="python $R/bench.py --steps 2 --warmup 1 --cpu-frames 0 --no-conv-events --no-extras --no-self-check ${BENCH_ARGS:-}"
TAG="${TAG:-}"
KERNEL="${KERNEL:-lwg_conv_winograd4_kernel}"
export LWG_PMC_CMD="bench.py --steps 2 --warmup 1 --no-extras --no-self-check ${BENCH_ARGS:-}"
run() { # name counters...
  local name=$1; shift
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$R/gpurun_out/pmc_$name$TAG" -o pmc -- $CMD > "$R/gpurun_out/pmc_$name$TAG.log" 2>&1 )
  echo "pmc $name$TAG exit=$?"
}
for p in ${PASSES:-A B C}; do
  case $p in
    A) run mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 ;;
    B) run fetch FETCH_SIZE ;;
    C) run write WRITE_SIZE ;;
    D) run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS ;;
    E) run l2 TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE ;;
    F) run l1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum GRBM_GUI_ACTIVE ;;
    G) run l2b TCC_READ_sum TCC_WRITE_sum TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum ;;
  esac
done
# per-launch HBM-side traffic of the dominant kernel -> profiles/pmc_traffic.json (read by bench.py) + per-kernel tables
if [ -d gpurun_out/pmc_fetch$TAG ] && [ -d gpurun_out/pmc_write$TAG ]; then
  python tools/pmc_summary.py --traffic gpurun_out/pmc_fetch$TAG gpurun_out/pmc_write$TAG gpurun_out/pmc_traffic$TAG.json "$KERNEL"
  python tools/pmc_summary.py gpurun_out/pmc_fetch$TAG gpurun_out/pmc_fetch$TAG.md > /dev/null 2>&1      # per-kernel tables (all kernels) before the raw CSVs are pruned
  python tools/pmc_summary.py gpurun_out/pmc_write$TAG gpurun_out/pmc_write$TAG.md > /dev/null 2>&1
  find gpurun_out/pmc_fetch$TAG gpurun_out/pmc_write$TAG -type f -size +3M -delete
fi
