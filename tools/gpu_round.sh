#!/bin/bash
# One GPU-box session: every parity check in its own process (a faulting kernel must not hide the others),
# then a short bench and (optionally) a rocprofv3 kernel trace.  Everything lands in gpurun_out/.
#   tools/gpu_round.sh [diag] [bench] [prof] [pytest]
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
WHAT="${*:-diag bench}"
{ /opt/rocm/bin/rocminfo | grep -E "Marketing Name|Compute Unit|gfx" | head -8; } > gpurun_out/device.txt 2>&1
if [[ " $WHAT " == *" diag "* ]]; then
  : > gpurun_out/diag_all.log
  for c in $(python - <<'PY'
import re
src = open("tests/gpu_checks.py").read()
print(" ".join(re.findall(r"^def (check_[a-z0-9_]+)\(", src, flags=re.M)))
PY
  ); do
    echo "=== $c" >> gpurun_out/diag_all.log
    timeout 600 python tools/gpu_diag.py "$c" >> gpurun_out/diag_all.log 2>&1
    echo "exit=$?" >> gpurun_out/diag_all.log
    cp gpurun_out/diag.json "gpurun_out/diag_$c.json" 2>/dev/null
  done
  grep -E "^(check_|=== |exit=|FAILED)" gpurun_out/diag_all.log | tail -60
fi
if [[ " $WHAT " == *" pytest "* ]]; then
  timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -25 | tee gpurun_out/pytest_gpu.log
fi
if [[ " $WHAT " == *" bench "* ]]; then
  timeout 900 python bench.py ${BENCH_ARGS:---steps 6 --warmup 2 --cpu-frames 2} > gpurun_out/bench.log 2> gpurun_out/bench.err
  echo "bench exit=$?"; tail -3 gpurun_out/bench.log; tail -5 gpurun_out/bench.err
fi
if [[ " $WHAT " == *" prof "* ]]; then
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof" -o lwg -- python "$OLDPWD/bench.py" --steps 4 --warmup 2 --cpu-frames 0 --no-conv-events --pipelined-streams 0 --output-frames 0 --no-split-extra > "$OLDPWD/gpurun_out/prof_bench.log" 2>&1 )
  echo "prof exit=$?"; find gpurun_out/prof -name "*stats*" | head; 
  f=$(find gpurun_out/prof -name "*kernel_stats*" | head -1); [ -n "$f" ] && head -25 "$f"
fi
