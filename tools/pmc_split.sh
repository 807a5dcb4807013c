#!/bin/bash
# PMC passes over the bf16x6 conv kernels alone (tools/convlab.py --split on one shape), PP = 0 and 1.
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
R="$PWD"
SHAPE=${SHAPE:-res64}
run() { # name pp counters...
  local name=$1 pp=$2; shift 2
  ( cd /tmp && LWG_SPLIT_PP=$pp timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$R/gpurun_out/pmcs_$name" -o pmc -- \
      python $R/tools/convlab.py --split --iters 5 --shapes $SHAPE $R/ipercore_amd/liblwg_hip.so > "$R/gpurun_out/pmcs_$name.log" 2>&1 )
  echo "pmc $name exit=$?"
  python tools/pmc_summary.py "gpurun_out/pmcs_$name" | grep -E "kernel|split" | cut -c1-400
}
for pp in 0 1; do
  run a$pp $pp SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
  run b$pp $pp SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD
done
