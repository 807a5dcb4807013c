#!/bin/bash
# round-2 evidence set (final code): full GPU suite, smoke, PMC traffic (fp32 512 / bf16 1024), rocprofv3 kernel stats of both,
# default bench, bf16 bench line, fp32 lines at 256 / 1024, personalization with the reference's default losses
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
R="$PWD"
echo "=== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
echo "=== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -2
echo "=== pmc fp32"; PASSES="B C" tools/pmc_round.sh 2>&1 | tail -1 | cut -c1-300
echo "=== pmc bf16"; PASSES="B C" TAG=_bf16 KERNEL=lwg_conv_bf16 BENCH_ARGS="--precision bf16 --size 1024 --workload novel_view" tools/pmc_round.sh 2>&1 | tail -1 | cut -c1-300
cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json 2>/dev/null; cp gpurun_out/pmc_traffic_bf16.json profiles/pmc_traffic_bf16.json 2>/dev/null
for cfg in "f32|" "bf16|--precision bf16 --size 1024 --workload novel_view"; do
  name=${cfg%%|*}; extra=${cfg#*|}
  echo "=== rocprof $name"; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_r02_$name" -o r02 -- python "$R/bench.py" --steps 3 --warmup 1 --no-extras --cpu-frames 0 --no-conv-events $extra > "$R/gpurun_out/prof_r02_$name.log" 2>&1 )
  f=$(find gpurun_out/prof_r02_$name -name "*kernel_stats*" | head -1); cp "$f" gpurun_out/r02_kernel_stats_$name.csv; python tools/prof_summary.py "$f" 4 30 > gpurun_out/r02_kernel_stats_$name.txt 2>&1; head -6 gpurun_out/r02_kernel_stats_$name.txt; tail -1 gpurun_out/r02_kernel_stats_$name.txt
  find gpurun_out/prof_r02_$name -type f -size +3M -delete
done
echo "=== default bench"; timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_r02_default.json 2> gpurun_out/bench_r02_default.err; echo "rc=$?"; python - <<'PY'
import json
d = json.loads(open('gpurun_out/bench_r02_default.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms/step', d['ms_per_step'], 'frac', d['roofline']['frac'], 'traffic', d['roofline']['traffic'], 'alg bytes', d['roofline']['algorithmic_bytes_per_launch'])
for k in ('pipelined', 'split_products', 'with_output', 'b1_latency', 'novel_view_1024_bf16', 'personalize_step', 'cpu_baseline'):
    v = d.get(k); print(k, json.dumps(v)[:330] if v else None)
PY
echo "=== bench bf16 1024"; timeout 600 python bench.py --precision bf16 --size 1024 --workload novel_view --steps 5 --warmup 2 --cpu-frames 0 > gpurun_out/bench_r02_bf16_1024.json 2>/dev/null; tail -1 gpurun_out/bench_r02_bf16_1024.json | cut -c1-400
for sz in 256 1024; do
  echo "=== bench fp32 $sz"; timeout 600 python bench.py --size $sz --steps 3 --warmup 1 --cpu-frames 0 --no-extras > gpurun_out/bench_r02_f32_$sz.json 2>/dev/null; tail -1 gpurun_out/bench_r02_f32_$sz.json | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('value', d['value'], 'fb', d['config']['frame_batch'], 'frac', d['roofline']['frac'])"
done
echo "=== personalize vgg+face"; timeout 600 python bench_personalize.py --steps 10 --warmup 3 --use-vgg --use-face > gpurun_out/bench_r02_personalize_vgg_face.json 2>/dev/null; tail -1 gpurun_out/bench_r02_personalize_vgg_face.json | cut -c1-300
echo "=== personalize"; timeout 600 python bench_personalize.py --steps 20 --warmup 5 > gpurun_out/bench_r02_personalize.json 2>/dev/null; tail -1 gpurun_out/bench_r02_personalize.json | cut -c1-300
