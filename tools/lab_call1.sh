cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x --durations=12 2>&1 | tail -40 > gpurun_out/pytest_gpu.log; echo "pytest exit=${PIPESTATUS[0]}"; tail -25 gpurun_out/pytest_gpu.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_a.json 2> gpurun_out/bench_a.err; echo "bench exit=$?"; head -c 1500 gpurun_out/bench_a.json; echo; tail -5 gpurun_out/bench_a.err
bash tools/final_profiles.sh prof 2>&1 | tail -5
head -40 gpurun_out/prof_summary.txt; head -40 gpurun_out/prof_bf16_summary.txt
