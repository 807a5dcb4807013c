cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
python tools/gpu_diag.py check_lwb_attention_x check_pipeline_tiny_64 check_benched_shapes_512 check_benched_shapes_1024_bf16 check_temporal_mode 2>&1 | grep -v Warning | tail -7
python tools/attnlab.py product product 2>&1 | tail -16
