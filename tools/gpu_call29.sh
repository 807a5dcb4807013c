#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
timeout 300 python - <<'PY' 2>&1 | tail -12
import sys, traceback
sys.path.insert(0, '.')
from tests import gpu_checks as g
try:
    r = g.check_lwb_attention()
    for k, v in r.items(): print(k, v)
except Exception as e:
    traceback.print_exc()
PY
