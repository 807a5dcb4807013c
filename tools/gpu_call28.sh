#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "=== parity"; timeout 900 python -m pytest tests -m gpu -x -q -k "attention or pipeline_full_512 or pipeline_full_256 or bf16_generator or num_source or temporal or lwb_variant or golden or edge" 2>&1 | tail -3
echo "=== kernel times (fp32 512 FB16, bf16 1024 FB8)"; timeout 600 python - <<'PY' 2>&1 | grep -v "Warning\|warn" | tail -14
import sys, time, torch
sys.path.insert(0, '.')
from ipercore_amd import synthetic as pu
from torch.profiler import profile, ProfilerActivity
for S, fb, prec, n in ((512, 16, "fp32", 64), (1024, 8, "bf16", 32)):
    case = pu.build_case(image_size=S, n_frames=n, ns=2)
    im = pu.make_imitator(case, frame_batch=fb, device=torch.device('cuda', 0))
    if prec != "fp32":
        im.generator.conv_precision = prec
        im.set_source(case.src_smpl, case.uv_img, case.bg_img, src_img=case.src_img)
    tgt = im.prepare_sequence(case.tgt_smpls, "smooth")
    for _ in range(2): im.synthesize(tgt, "smooth")
    torch.cuda.synchronize()
    t0 = time.perf_counter(); im.synthesize(tgt, "smooth"); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        im.synthesize(tgt, "smooth"); torch.cuda.synchronize()
    tot = sum(e.device_time_total for e in prof.key_averages())
    print(S, prec, "fps", round(n / dt, 1), "kernel ms", round(tot / 1e3, 2))
    for e in prof.key_averages():
        if any(k in e.key for k in ("attn", "flow_resize")): print("   ", e.key[:44], e.count, round(e.device_time_total / e.count, 1), "us avg", round(100 * e.device_time_total / tot, 2), "%")
PY
