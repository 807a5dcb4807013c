#!/bin/bash
# Round-6 call 10: the activation resolved once per workgroup in every convolution epilogue - parity, then A/B against HEAD's library (fp32 512 headline, bf16 1024)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 1500 python tools/gpu_diag.py check_winograd_mode check_winograd_up4 check_conv_variants check_conv_transpose check_spade_epilogue check_bf16_generator check_benched_shapes_512 check_benched_shapes_1024_bf16 check_generator_golden 2>&1 | grep -v amdgpu.ids | tail -12
cp $O/diag.json $O/r06_j_act_dispatch_checks.json
bash tools/ab_bench.sh tools/lab/liblwg_head.so --steps 8 --warmup 4 2>&1 | tee $O/r06_j_ab_f32_512.txt
bash tools/ab_bench.sh tools/lab/liblwg_head.so --precision bf16 --size 1024 --workload novel_view --steps 4 --warmup 2 2>&1 | tee $O/r06_j_ab_bf16_1024.txt
