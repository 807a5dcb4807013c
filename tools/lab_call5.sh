cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
python tools/attnx_debug.py 2>&1 | tail -11
python tools/attnlab.py product tools/lab/liblwg_attnx_occ5.so tools/lab/liblwg_attnx_occ6.so 2>&1 | tail -24
python tools/gpu_diag.py check_lwb_attention_x check_pipeline_tiny_64 check_benched_shapes_512 check_num_source_1_and_8 check_temporal_mode check_batch_slicing_1024 check_benched_shapes_1024_bf16 2>&1 | grep -v Warning | tail -9
timeout 600 python tools/diag_train512.py 2>&1 | grep -v Warning | tail -19
