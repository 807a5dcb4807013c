#!/bin/bash
# column-keyed halo swizzle (lab + PMC conflicts + bf16 parity), rasterizer with 32-pixel bins (parity + kernel time), personalization graph + panel cache
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
R="$PWD"
echo "=== bf16lab HR batch x4 (column-keyed swizzle)"; timeout 600 python tools/bf16lab.py --no-f32 --convs-only --batch-mul 4 2>&1 | tee gpurun_out/bf16lab_v7_colswz_bm4.txt | grep -v amdgpu.ids
C="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$R/gpurun_out/pmc_bf16_hr_v7" -o pmc -- python "$R/tools/bf16lab.py" --no-f32 --convs-only --batch-mul 4 --shapes res64,skip0,up2 --iters 4 > "$R/gpurun_out/pmc_bf16_hr_v7.log" 2>&1 )
python tools/pmc_summary.py gpurun_out/pmc_bf16_hr_v7 gpurun_out/pmc_bf16_hr_v7.md 2>/dev/null | grep -E "^\| kernel|lwg_conv_bf16" | cut -c1-500
echo "=== bf16 checks + raster"; timeout 900 python - <<'PY' 2>&1 | tail -8
import sys, json, time
sys.path.insert(0, '.')
from tests import gpu_checks as g
for name in ("check_bf16_generator", "check_bf16_vs_oracle", "check_raster", "check_pipeline_full_512"):
    t0 = time.time()
    try:
        r = getattr(g, name)()
        print(name, "OK", round(time.time() - t0, 1), "s", json.dumps(r, default=str)[:500], flush=True)
    except Exception as e:
        import traceback; traceback.print_exc()
        print(name, "FAILED", type(e).__name__, str(e)[:800], flush=True)
PY
echo "=== bench bf16 1024"; timeout 600 python bench.py --precision bf16 --size 1024 --workload novel_view --steps 3 --warmup 1 --no-extras --cpu-frames 0 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('value', d['value'], 'ms/step', d['ms_per_step'], 'fb', d['config']['frame_batch'], 'conv TF', r['achieved'], 'share', r['share_of_step_time'], 'gov', r.get('frac_of_governing_roof'))"
for v in "" "--wgrad-side" "--no-panel-cache"; do
  echo "=== personalize $v"; timeout 600 python bench_personalize.py --steps 10 --warmup 3 $v 2>gpurun_out/pers.err | tail -1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); print('ms/step', d['ms_per_step'], 'TF', d['conv_tflops_whole_step'], 'host', d['single_step_host_enqueue_ms'], d['config']['step'][:30], 'loss', d['loss_G'], d['loss_D'])
except Exception as e: print('FAILED', e)"; grep -A8 "Raised at" gpurun_out/pers.err | head -12
done
echo "=== rocprof fp32 short"; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_r02_f32b" -o r02 -- python "$R/bench.py" --steps 2 --warmup 1 --no-extras --cpu-frames 0 --no-conv-events > "$R/gpurun_out/prof_r02_f32b.log" 2>&1 ); tail -1 gpurun_out/prof_r02_f32b.log | cut -c1-120
f=$(find gpurun_out/prof_r02_f32b -name "*kernel_stats*" | head -1); python tools/prof_summary.py "$f" 3 12 2>&1 | grep -E "raster|head|attn|total"
find gpurun_out/prof_r02_f32b -type f -size +4M -delete
