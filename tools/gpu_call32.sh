#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for occ in 4 5 6; do
  echo "=== bf16 OCC=$occ"; LWG_ATTN16_OCC=$occ timeout 600 python - <<'PY' 2>&1 | grep -v "Warning\|warn\|amdgpu" | tail -4
import sys, time, torch
sys.path.insert(0, '.')
from ipercore_amd import synthetic as pu
from torch.profiler import profile, ProfilerActivity
S, fb, n = 1024, 8, 32
case = pu.build_case(image_size=S, n_frames=n, ns=2)
im = pu.make_imitator(case, frame_batch=fb, device=torch.device('cuda', 0))
im.generator.conv_precision = "bf16"
im.set_source(case.src_smpl, case.uv_img, case.bg_img, src_img=case.src_img)
tgt = im.prepare_sequence(case.tgt_smpls, "smooth")
for _ in range(3): im.synthesize(tgt, "smooth")
torch.cuda.synchronize()
t0 = time.perf_counter(); im.synthesize(tgt, "smooth"); torch.cuda.synchronize(); dt = time.perf_counter() - t0
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    im.synthesize(tgt, "smooth"); torch.cuda.synchronize()
tot = sum(e.device_time_total for e in prof.key_averages())
print("fps", round(n / dt, 1), "kernel ms", round(tot / 1e3, 2))
for e in prof.key_averages():
    if "attn" in e.key: print("   ", e.key[:52], e.count, round(e.device_time_total / e.count, 1), "us avg", round(100 * e.device_time_total / tot, 2), "%")
PY
done
for occ in 5 6 8; do
  echo "=== fp32 OCC=$occ"; LWG_ATTN_OCC=$occ timeout 600 python - <<'PY' 2>&1 | grep -v "Warning\|warn\|amdgpu" | tail -4
import sys, time, torch
sys.path.insert(0, '.')
from ipercore_amd import synthetic as pu
from torch.profiler import profile, ProfilerActivity
S, fb, n = 512, 16, 64
case = pu.build_case(image_size=S, n_frames=n, ns=2)
im = pu.make_imitator(case, frame_batch=fb, device=torch.device('cuda', 0))
tgt = im.prepare_sequence(case.tgt_smpls, "smooth")
for _ in range(3): im.synthesize(tgt, "smooth")
torch.cuda.synchronize()
t0 = time.perf_counter(); im.synthesize(tgt, "smooth"); torch.cuda.synchronize(); dt = time.perf_counter() - t0
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    im.synthesize(tgt, "smooth"); torch.cuda.synchronize()
tot = sum(e.device_time_total for e in prof.key_averages())
print("fps", round(n / dt, 1), "kernel ms", round(tot / 1e3, 2))
for e in prof.key_averages():
    if "attn" in e.key: print("   ", e.key[:52], e.count, round(e.device_time_total / e.count, 1), "us avg", round(100 * e.device_time_total / tot, 2), "%")
PY
done
