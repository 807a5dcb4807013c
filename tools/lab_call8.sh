cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
python tools/gpu_diag.py check_lwb_attention_x check_pipeline_tiny_64 check_benched_shapes_512 2>&1 | grep -v Warning | tail -5
python tools/attnlab.py product tools/lab/liblwg_attnx_bvreg.so tools/lab/liblwg_attnx_occ16_3.so tools/lab/liblwg_attnx_nowide.so product 2>&1 | tail -40
