cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
python tools/attnx_debug.py 2>&1 | tail -12
timeout 1800 python -m pytest tests -q -m gpu --durations=8 > gpurun_out/pytest_gpu_full.log 2>&1; echo "pytest exit=$?"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/pytest_gpu_full.log | tail -30
grep -E "^E  " gpurun_out/pytest_gpu_full.log | head -40
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_b.json 2> gpurun_out/bench_b.err; echo "bench exit=$?"; head -c 600 gpurun_out/bench_b.json; echo; tail -3 gpurun_out/bench_b.err
bash tools/final_profiles.sh prof 2>&1 | tail -3
head -16 gpurun_out/prof_summary.txt; head -20 gpurun_out/prof_bf16_summary.txt
