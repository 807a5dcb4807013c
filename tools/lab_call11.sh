cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
python tools/up4lab.py tools/lab/liblwg_up4_nopair.so product tools/lab/liblwg_up4_nopair.so product 2>&1 | tail -20
python tools/gpu_diag.py check_bf16_conv_kernels check_benched_shapes_1024_bf16 check_bf16_generator 2>&1 | grep -v Warning | tail -5
