#!/bin/bash
# round-2 GPU call 1: probes, the widened parity suite, the new bench line
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
./build/dma_probe > gpurun_out/dma_probe.txt 2>&1; cat gpurun_out/dma_probe.txt
timeout 1500 python -m pytest tests -m gpu -x -q --durations=25 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -40 gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 4 --warmup 2 > gpurun_out/bench_call1.json 2> gpurun_out/bench_call1.err; echo "bench exit $?"; tail -c 6000 gpurun_out/bench_call1.json; tail -5 gpurun_out/bench_call1.err
