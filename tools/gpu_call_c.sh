#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O/suite
timeout 1200 python -m pytest tests -q -m gpu -k "check_face_loss or check_discriminator_and_trainer_step or check_graph_vs_eager" 2>&1 | tail -25 > $O/suite/pytest_c.log; echo "pytest exit=${PIPESTATUS[0]}"; tail -6 $O/suite/pytest_c.log
timeout 600 python bench_personalize.py --steps 10 --warmup 4 --use-vgg --use-face > $O/pers_vgg_face.json 2> $O/pers_vgg_face.err; echo "pers vgg face exit=$?"; python - <<'PY'
import json
for f in ("pers_vgg_face",):
    d=json.loads([l for l in open(f"gpurun_out/{f}.json") if l.startswith("{")][-1])
    print(f, d["ms_per_step"], d["roofline"]["frac"], d["config"]["step"][:60], d.get("self_check"), d.get("self_check_detail"), d["single_step_host_enqueue_ms"])
PY
tail -3 $O/pers_vgg_face.err | cut -c1-300
timeout 600 python bench_personalize.py --steps 10 --warmup 4 > $O/pers.json 2> $O/pers.err; echo "pers exit=$?"; python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/pers.json") if l.startswith("{")][-1])
print("pers", d["ms_per_step"], d["roofline"]["frac"], d.get("self_check"), d.get("self_check_detail"))
PY
