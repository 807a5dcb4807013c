#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/suite; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 2400 python -m pytest tests -q -m gpu --durations=6 2>&1 | tail -25 > $O/pytest_gpu_full.log; echo "pytest exit=${PIPESTATUS[0]}"; tail -14 $O/pytest_gpu_full.log
bash tools/prof_pers.sh > $O/prof_pers.txt 2>&1; head -12 $O/prof_pers.txt
bash tools/prof_pers.sh --use-vgg --use-face > $O/prof_pers_vgg_face.txt 2>&1; head -12 $O/prof_pers_vgg_face.txt
