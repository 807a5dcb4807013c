#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 1500 python tools/gpu_diag.py check_bf16_up4_head check_bf16_generator check_bf16_vs_oracle check_benched_shapes_1024_bf16 2>&1 | grep -v amdgpu.ids | tail -8
cp $O/diag.json $O/r06_d_bf16_fused_checks.json 2>/dev/null
timeout 1500 python tools/bf16_ab.py 4 2>&1 | grep -v amdgpu.ids | tee $O/r06_d_bf16_up4_head_ab.txt
