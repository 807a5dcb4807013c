#!/bin/bash
# Round-5 lab call 1: Winograd kernel variants - correctness, per-shape times, phase timeline, PMC (all under gpurun_out/w1)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
R=$PWD; O=gpurun_out/w1; mkdir -p $O; L=tools/lab
timeout 300 python tools/winolab.py $L/liblwg_w_base.so $L/liblwg_w_fine4.so $L/liblwg_w_epi4.so $L/liblwg_w_fine.so > $O/winolab.log 2>&1; echo "winolab exit=$?"
for v in base fine4; do timeout 200 python tools/winoshapes.py --lib $L/liblwg_w_$v.so > $O/shapes_$v.log 2>&1; echo "shapes $v exit=$?"; done
for v in vs96 prio f128 fine epi4; do timeout 200 python tools/winoshapes.py --lib $L/liblwg_w_$v.so --nodirect > $O/shapes_$v.log 2>&1; echo "shapes $v exit=$?"; done
for v in ts finets; do for i in 0 1 3 5 8; do timeout 100 python tools/winoshapes.py --lib $L/liblwg_w_$v.so --ts --only $i; done > $O/ts_$v.log 2>&1; done
pmc() { # name lib shape counters...
  local name=$1 lib=$2 shape=$3; shift 3
  ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$R/$O/pmc_$name" -o pmc -- python $R/tools/winoshapes.py --lib $R/$L/liblwg_w_$lib.so --only $shape --nodirect --reps 5 > "$R/$O/pmc_$name.log" 2>&1 )
  python tools/pmc_summary.py $O/pmc_$name $O/pmc_$name.md > /dev/null 2>&1
  find $O/pmc_$name -type f -size +1M -delete
}
for lib in base fine4; do for sh in 0 5; do
  pmc ${lib}_s${sh}_mfma $lib $sh SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU
  pmc ${lib}_s${sh}_lds $lib $sh SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE
done; done
tail -n 30 $O/winolab.log $O/shapes_*.log $O/ts_*.log
cat $O/pmc_*.md | grep -v "^|---" | grep -i "wino\|kernel |"
