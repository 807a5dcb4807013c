#!/bin/bash
# Round-5 end-of-round evidence (one gpurun call): the -m gpu suite, the driver's bench command, rocprofv3 kernel stats + PMC passes of the default
# engine (fp32 512^2 clip) and of the bf16 1024^2 run, PMC of the Winograd kernel on two launch shapes, the personalization kernel stats.
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
R=$PWD; O=gpurun_out; mkdir -p $O
bash tools/final_profiles.sh pytest bench prof pmc 2>&1 | tail -30
pmc() { # name shape counters...
  local name=$1 shape=$2; shift 2
  ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$R/$O/pmc_wino_$name" -o pmc -- python $R/tools/winoshapes.py --only $shape --nodirect --reps 5 --frames 64 > "$R/$O/pmc_wino_$name.log" 2>&1 )
  python tools/pmc_summary.py $O/pmc_wino_$name $O/pmc_wino_$name.md > /dev/null 2>&1
  find $O/pmc_wino_$name -type f -size +1M -delete
}
for sh in ${WINO_PMC_SHAPES-0 5 7}; do          # WINO_PMC_SHAPES="" skips the per-shape PMC passes (the kernel's whole-launch form did not change)
  pmc s${sh}_mfma $sh SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU
  pmc s${sh}_lds $sh SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE
done
cat $O/pmc_wino_*.md | grep -i "wino\|kernel |" | cut -c1-330
timeout 300 python tools/winoshapes.py --frames 64 > $O/winoshapes_64.log 2>&1; grep -v amdgpu $O/winoshapes_64.log | tail -11
bash tools/prof_pers.sh > $O/prof_pers_final.txt 2>&1; head -8 $O/prof_pers_final.txt
bash tools/prof_pers.sh --use-vgg --use-face > $O/prof_pers_vgg_face_final.txt 2>&1; head -6 $O/prof_pers_vgg_face_final.txt
timeout 300 python tools/winoshapes.py --frames 1 --splitk --vgg 2>&1 | grep -v amdgpu > $O/winoshapes_split_vgg.log
timeout 300 python tools/winoshapes.py --frames 1 --splitk 2>&1 | grep -v amdgpu > $O/winoshapes_split_f1.log; tail -3 $O/winoshapes_split_f1.log | cut -c1-200
