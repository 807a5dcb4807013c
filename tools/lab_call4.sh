cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python tools/diag_train512.py 2>&1 | grep -v Warning | tail -12
