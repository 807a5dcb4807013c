#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "=== default bench (the driver's command)"; timeout 1500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_r02_default.json 2> gpurun_out/bench_r02_default.err; echo "rc=$?"; python - <<'PY'
import json
d = json.loads(open('gpurun_out/bench_r02_default.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms/step', d['ms_per_step'], 'frac', d['roofline']['frac'], 'traffic', d['roofline']['traffic'], 'fb', d['config']['frame_batch'])
for k in ('pipelined', 'with_output', 'novel_view_1024_bf16', 'personalize_step'):
    v = d.get(k); print(k, json.dumps(v)[:200] if v else None)
PY
