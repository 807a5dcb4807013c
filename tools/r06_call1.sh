#!/bin/bash
# Round-6 call 1: the adversarial-distribution parity check of both Winograd kernels + the changed checks, the bisect of the
# background network's gradient outliers, PMC passes (MFMA busy, FETCH / WRITE per kernel family at the bench's real frame batch, LDS bank
# conflicts) of the default engine, PMC of the personalization step.
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 1500 python tools/gpu_diag.py check_winograd_adversarial check_panel_cache_refresh check_face_loss 2>&1 | grep -v amdgpu.ids | tail -8
cp $O/diag.json $O/r06_a_new_checks.json 2>/dev/null
timeout 1500 python tools/diag_bg_grads.py 512 4 2>&1 | grep -v amdgpu.ids > $O/r06_a_bg_grads_bisect.txt; tail -45 $O/r06_a_bg_grads_bisect.txt
PASSES="A B C D" bash tools/pmc_round.sh 2>&1 | tail -8
for p in mfma lds; do python tools/pmc_summary.py $O/pmc_$p $O/pmc_$p.md > /dev/null 2>&1; done
grep -i "wino\|igemm\|kernel |" $O/pmc_mfma.md $O/pmc_lds.md | cut -c1-400
bash tools/pmc_pers.sh 2>&1 | tail -14
find $O -type f -size +3M -delete
