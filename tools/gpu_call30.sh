#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
echo "=== pytest gpu (full)"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
