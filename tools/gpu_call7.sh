#!/bin/bash
# Round-2 validation call: N=64 HR lab A/B, full GPU suite, smoke, default bench, rocprofv3 kernel stats of the default bench.
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "=== bf16lab HR (N=64 included) batch x4"; timeout 600 python tools/bf16lab.py --no-f32 --convs-only --batch-mul 4 2>&1 | tee gpurun_out/bf16lab_v5_hr64_bm4.txt | grep -v amdgpu.ids
echo "=== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
echo "=== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -3
echo "=== default bench"; timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_r02_default.json 2> gpurun_out/bench_r02_default.err; tail -c 6000 gpurun_out/bench_r02_default.json
echo "=== rocprof headline"; cd /tmp; timeout 900 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_r02_f32" -o r02_f32 -- python "$GRAFT_REPO_ROOT/bench.py" --steps 20 --warmup 5 --no-extras --cpu-frames 0 > "$GRAFT_REPO_ROOT/gpurun_out/prof_r02_f32.log" 2>&1; tail -1 "$GRAFT_REPO_ROOT/gpurun_out/prof_r02_f32.log" | cut -c1-600
cd "$GRAFT_REPO_ROOT"; find gpurun_out/prof_r02_f32 -name "*kernel_stats*" | head; find gpurun_out/prof_r02_f32 -type f ! -name "*stats*" -size +2M -delete
