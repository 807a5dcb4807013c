cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
python tools/attnlab.py tools/lab/liblwg_attnx_v1.so tools/lab/liblwg_attnx_nsu.so product tools/lab/liblwg_attnx_v1occ6.so tools/lab/liblwg_attnx_v1.so 2>&1 | tail -40
timeout 600 python tools/diag_train512.py 2>&1 | grep -v Warning | tail -28
