cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -q -m gpu --durations=8 > gpurun_out/pytest_gpu_full.log 2>&1; echo "pytest exit=$?"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/pytest_gpu_full.log | tail -20
grep -E "^E  " gpurun_out/pytest_gpu_full.log | head -30
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_c.json 2> gpurun_out/bench_c.err; echo "bench exit=$?"; head -c 400 gpurun_out/bench_c.json; echo; tail -3 gpurun_out/bench_c.err
