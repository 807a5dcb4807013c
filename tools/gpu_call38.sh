#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python bench.py --steps 4 --warmup 2 --no-extras --cpu-frames 0 --conv-breakdown 2>/dev/null | tail -1 | cut -c1-200
python - <<'PY'
import json
rows = json.load(open('gpurun_out/conv_breakdown.json'))
tot = sum(r['ms'] for r in rows)
for r in sorted(rows, key=lambda r: -r['ms'])[:24]:
    print(f"{r['shape']:46s} n={r['launches']:5d} us={1000*r['ms']/r['launches']:8.1f} share={100*r['ms']/tot:5.1f}% TF={r['tflops']:.1f} frac={r['tflops']/157.3:.3f}")
PY
