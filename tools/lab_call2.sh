cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
python tools/attnx_debug.py 2>&1 | tail -20
timeout 1500 python -m pytest tests -q -m gpu --durations=12 2>&1 | tail -60 > gpurun_out/pytest_gpu.log; echo "pytest exit=${PIPESTATUS[0]}"; tail -45 gpurun_out/pytest_gpu.log
