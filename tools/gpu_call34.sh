#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
echo "=== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -1
echo "=== personalize"; timeout 600 python bench_personalize.py --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('ms/step', d['ms_per_step'], d['config']['step'][:50], 'exposed', d.get('exposed_allreduce_ms_per_step'))"
echo "=== trainer tests"; timeout 900 python -m pytest tests -m gpu -x -q -k "train or personal" 2>&1 | tail -2
