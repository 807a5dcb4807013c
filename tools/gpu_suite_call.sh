#!/bin/bash
# usage: gpu_suite_call.sh [-k EXPR]   the -m gpu suite (optionally a subset), logs under gpurun_out/suite
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/suite; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu --durations=8 "$@" 2>&1 | tail -60 > $O/pytest_gpu.log; echo "pytest exit=${PIPESTATUS[0]}"; tail -25 $O/pytest_gpu.log
