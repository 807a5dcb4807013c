#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
R="$PWD"
for bm in 1 2 4; do
  echo "=== bf16lab batch x$bm"; timeout 600 python tools/bf16lab.py --no-f32 --convs-only --batch-mul $bm 2>&1 | tee gpurun_out/bf16lab_v2_bm$bm.txt | grep -v amdgpu.ids
done
echo "=== bf16lab batch x2 BIG=0"; LWG_BF16_BIG=0 timeout 600 python tools/bf16lab.py --no-f32 --convs-only --batch-mul 2 --shapes res64,gb64,skip0,up0 2>&1 | tee gpurun_out/bf16lab_v2_bm2_nobig.txt | grep -v amdgpu.ids
echo "=== pmc"
for big in 0 1; do
 for pass in A B; do
  if [ $pass = A ]; then C="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"; else C="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS"; fi
  ( cd /tmp && LWG_BF16_BIG=$big timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$R/gpurun_out/pmc_bf16_big${big}_$pass" -o pmc -- python "$R/tools/bf16lab.py" --no-f32 --convs-only --batch-mul 2 --shapes res64,skip0 --iters 4 > "$R/gpurun_out/pmc_bf16_big${big}_$pass.log" 2>&1 )
  python tools/pmc_summary.py gpurun_out/pmc_bf16_big${big}_$pass gpurun_out/pmc_bf16_big${big}_$pass.md 2>/dev/null | grep -E "kernel|lwg_conv_bf16" | cut -c1-400
 done
done
for fb in 2 4 8; do
  echo "=== bench bf16 1024 fb=$fb"; timeout 600 python bench.py --precision bf16 --size 1024 --workload novel_view --steps 3 --warmup 1 --no-extras --cpu-frames 0 --frame-batch $fb 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('value', d['value'], 'ms/step', d['ms_per_step'], 'conv TF', r['achieved'], 'share', r['share_of_step_time'], 'avg us', r['avg_launch_us'])"
done
echo "=== bench bf16 1024 fb=4 streams=3"; timeout 600 python bench.py --precision bf16 --size 1024 --workload novel_view --steps 3 --warmup 1 --no-extras --cpu-frames 0 --frame-batch 4 --streams 3 --no-conv-events 2>&1 | tail -1 | cut -c1-200
echo "=== personalize graph"; timeout 600 python bench_personalize.py --steps 10 --warmup 3 2>&1 | tail -2 | cut -c1-1500
echo "=== personalize eager"; timeout 600 python bench_personalize.py --steps 10 --warmup 3 --no-graph 2>&1 | tail -1 | cut -c1-600
echo "=== trainer checks"; timeout 900 python - <<'PY' 2>&1 | tail -12
import sys, json, time
sys.path.insert(0, '.')
from tests import gpu_checks as g
for name in ("check_train_ops", "check_discriminator_and_trainer_step", "check_personalize_loop", "check_vgg_loss", "check_conv_backward"):
    t0 = time.time()
    try:
        r = getattr(g, name)()
        print(name, "OK", round(time.time() - t0, 1), "s", flush=True)
    except Exception as e:
        import traceback; traceback.print_exc()
        print(name, "FAILED", type(e).__name__, str(e)[:800], flush=True)
PY
