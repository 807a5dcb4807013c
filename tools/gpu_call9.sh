#!/bin/bash
# personalization step: graph capture after an eager step (fix), batched panel packing, side-stream wgrad, split precision; trainer checks;
# PMC passes on the register-streamed-weights bf16 conv kernel
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
R="$PWD"
echo "=== graph crash diag []"; timeout 300 python -X faulthandler tools/diag_graph_crash.py 2>&1 | grep -v "amdgpu.ids" | tail -4 | cut -c1-400
echo "=== graph crash diag [pipelined,output]"; timeout 300 python -X faulthandler tools/diag_graph_crash.py pipelined,output 2>&1 | grep -v "amdgpu.ids" | tail -3 | cut -c1-400
for v in "" "--no-panel-cache" "--wgrad-side" "--precision split" "--precision split --wgrad-side" "--no-graph"; do
  echo "=== personalize $v"; timeout 600 python bench_personalize.py --steps 10 --warmup 3 $v 2>/dev/null | tail -1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); print('ms/step', d['ms_per_step'], 'TF', d['conv_tflops_whole_step'], 'host', d['single_step_host_enqueue_ms'], d['config']['step'][:30], 'loss', d['loss_G'], d['loss_D'])
except Exception as e: print('FAILED', e)"
done
echo "=== trainer checks"; timeout 900 python - <<'PY' 2>&1 | tail -12
import sys, json, time
sys.path.insert(0, '.')
from tests import gpu_checks as g
for name in ("check_train_ops", "check_discriminator_and_trainer_step", "check_personalize_loop", "check_vgg_loss", "check_conv_backward", "check_generator_training_grads"):
    if not hasattr(g, name):
        print(name, "absent"); continue
    t0 = time.time()
    try:
        r = getattr(g, name)()
        print(name, "OK", round(time.time() - t0, 1), "s", flush=True)
    except Exception as e:
        import traceback; traceback.print_exc()
        print(name, "FAILED", type(e).__name__, str(e)[:800], flush=True)
PY
echo "=== pmc HR kernel"
for pass in A B; do
  if [ $pass = A ]; then C="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"; else C="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU"; fi
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$R/gpurun_out/pmc_bf16_hr_$pass" -o pmc -- python "$R/tools/bf16lab.py" --no-f32 --convs-only --batch-mul 4 --shapes res64,skip0,up2 --iters 4 > "$R/gpurun_out/pmc_bf16_hr_$pass.log" 2>&1 )
  python tools/pmc_summary.py gpurun_out/pmc_bf16_hr_$pass gpurun_out/pmc_bf16_hr_$pass.md 2>/dev/null | grep -E "^\| kernel|lwg_conv_bf16" | cut -c1-500
done
