cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
python tools/attnlab.py tools/lab/liblwg_attnx_nsu.so tools/lab/liblwg_attnx_nsui.so tools/lab/liblwg_attnx_v1i.so tools/lab/liblwg_attnx_nsu.so tools/lab/liblwg_attnx_nsui.so 2>&1 | tail -40
python tools/gpu_diag.py check_generator_training_grads_512_full 2>&1 | grep -v Warning | tail -4
