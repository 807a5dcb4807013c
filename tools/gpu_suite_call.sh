#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/suite; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x --durations=8 2>&1 | tail -40 > $O/pytest_gpu.log; echo "pytest exit=${PIPESTATUS[0]}"; tail -15 $O/pytest_gpu.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench exit=$?"; head -c 2500 $O/bench.json; echo; tail -3 $O/bench.err
