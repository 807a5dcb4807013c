#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O/suite
timeout 900 python -m pytest tests -q -m gpu -k "check_generator_golden_256 or check_generator_training_grads_512_full or check_winograd_mode" 2>&1 | tail -15 > $O/suite/pytest_new.log; echo "pytest exit=${PIPESTATUS[0]}"; tail -6 $O/suite/pytest_new.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_b.json 2> $O/bench_b.err; echo "bench exit=$?"; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_b.json'))
print(d["value"], d["self_check"], d["roofline"]["frac"], d["roofline"].get("winograd_kernel_frac"))
print(json.dumps(d.get("shard_of_8"))[:1500])
PY
bash tools/final_profiles.sh prof pmc 2>&1 | tail -12
