#!/bin/bash
# Round-6 call 11: one-input / two-input forms of the Winograd kernel (no per-load source branch) - parity, A/B against HEAD's library
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 1500 python tools/gpu_diag.py check_winograd_mode check_winograd_adversarial check_benched_shapes_512 check_generator_golden check_pipeline_full_512 check_generator_training_grads 2>&1 | grep -v amdgpu.ids | tail -8
cp $O/diag.json $O/r06_l_two_input_checks.json
bash tools/ab_bench.sh tools/lab/liblwg_head.so --steps 8 --warmup 4 2>&1 | tee $O/r06_l_ab_f32_512.txt
