#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "=== lds throughput probe"; timeout 60 build/lds_tput_probe 2>&1 | tee gpurun_out/lds_tput_probe.txt
for v in "" "--branch-streams" "--branch-streams --wgrad-side"; do
  echo "=== personalize $v"; timeout 600 python bench_personalize.py --steps 10 --warmup 3 $v 2>gpurun_out/pers.err | tail -1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); print('ms/step', d['ms_per_step'], 'TF', d['conv_tflops_whole_step'], 'host', d['single_step_host_enqueue_ms'], d['config']['step'][:30], 'loss', d['loss_G'], d['loss_D'])
except Exception as e: print('FAILED', e)"; grep -A8 "Raised at" gpurun_out/pers.err | head -12
done
echo "=== raster time"; timeout 300 python - <<'PY' 2>&1 | tail -3
import sys, time, torch
sys.path.insert(0, '.')
from ipercore_amd import synthetic as pu
case = pu.build_case(image_size=512, n_frames=64, ns=2)
im = pu.make_imitator(case, frame_batch=16, device=torch.device('cuda', 0))
tgt = im.prepare_sequence(case.tgt_smpls, "smooth")
for _ in range(2): im.synthesize(tgt, "smooth")
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    im.synthesize(tgt, "smooth"); torch.cuda.synchronize()
for e in prof.key_averages():
    if "raster" in e.key or "head_compose" in e.key: print(e.key[:60], e.count, round(e.device_time_total / e.count, 1), "us avg")
PY
