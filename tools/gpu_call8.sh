#!/bin/bash
# pointwise / first-layer bf16 kernels (lab + whole path), the in-process graph-capture crash, default bench, rocprofv3 csv stats
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "=== bf16lab new kernels"; timeout 300 python tools/bf16lab.py --no-f32 --convs-only --batch-mul 4 --shapes fq64,fq128,fq256,enc0,up2,enc1 2>&1 | grep -v amdgpu.ids | tee gpurun_out/bf16lab_v6_pw_c8.txt
echo "=== same, old paths"; LWG_LAB_PW=0 LWG_LAB_C8=0 timeout 300 python tools/bf16lab.py --no-f32 --convs-only --batch-mul 4 --shapes fq64,fq128,fq256,enc0 2>&1 | grep -v amdgpu.ids | tee gpurun_out/bf16lab_v6_old.txt
echo "=== bf16 checks"; timeout 900 python - <<'PY' 2>&1 | tail -8
import sys, json, time
sys.path.insert(0, '.')
from tests import gpu_checks as g
for name in ("check_bf16_generator", "check_bf16_vs_oracle"):
    t0 = time.time()
    try:
        r = getattr(g, name)()
        print(name, "OK", round(time.time() - t0, 1), "s", json.dumps(r, default=str)[:900], flush=True)
    except Exception as e:
        import traceback; traceback.print_exc()
        print(name, "FAILED", type(e).__name__, str(e)[:800], flush=True)
PY
echo "=== bench bf16 1024"; timeout 600 python bench.py --precision bf16 --size 1024 --workload novel_view --steps 3 --warmup 1 --no-extras --cpu-frames 0 --conv-breakdown 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('value', d['value'], 'ms/step', d['ms_per_step'], 'fb', d['config']['frame_batch'], 'conv TF', r['achieved'], 'share', r['share_of_step_time'], 'gov', r.get('frac_of_governing_roof'))"
cp gpurun_out/conv_breakdown.json gpurun_out/conv_breakdown_bf16_1024_v6.json
for st in "" pipelined output split,b1 bf16; do
  echo "=== graph crash diag: [$st]"; timeout 300 python -X faulthandler tools/diag_graph_crash.py $st 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" | tail -12
done
echo "=== default bench"; timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_r02_default.json 2> gpurun_out/bench_r02_default.err; echo "rc=$?"; tail -c 3000 gpurun_out/bench_r02_default.json
