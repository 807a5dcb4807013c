#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_ar_determinism2.txt; : > $O
for v in ctw_linear; do
  l=""; [ $v != tree ] && l="--lib tools/lab/liblwg_$v.so"
  echo "== $v" >> $O
  timeout 600 python tools/determinism_stress.py $l --reps 12 2>&1 | grep -v amdgpu.ids | tail -12 >> $O
done
cat $O
