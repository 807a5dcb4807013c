#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
echo "=== diag batch invariance"; timeout 300 python tools/diag_batch_invariance.py 512 8 2>&1 | tail -12
echo "=== bf16lab DMA_A=1"; LWG_BF16_DMA_A=1 timeout 600 python tools/bf16lab.py --no-f32 2>&1 | tee gpurun_out/bf16lab_dma1.txt | tail -40
echo "=== bf16lab TILE64"; LWG_BF16_TILE64=1 timeout 600 python tools/bf16lab.py --no-f32 --shapes res64,skip0,skip1,up1,up0 2>&1 | tee gpurun_out/bf16lab_tile64.txt | head -8
echo "=== bf16 checks"; timeout 900 python - <<'PY' 2>&1 | tail -30
import sys, json, time
sys.path.insert(0, '.')
from tests import gpu_checks as g
for name in ("check_bf16_generator", "check_novel_view_256", "check_num_source_1_and_8", "check_pipeline_full_1024", "check_bf16_vs_oracle"):
    t0 = time.time()
    try:
        r = getattr(g, name)()
        print(name, "OK", round(time.time() - t0, 1), "s", json.dumps(r, default=str)[:1500], flush=True)
    except Exception as e:
        import traceback; traceback.print_exc()
        print(name, "FAILED", type(e).__name__, str(e)[:1500], flush=True)
PY
echo "=== bench bf16 1024"; timeout 600 python bench.py --precision bf16 --size 1024 --workload novel_view --steps 3 --warmup 1 --no-extras --cpu-frames 0 --conv-breakdown 2>&1 | tail -3
cp gpurun_out/conv_breakdown.json gpurun_out/conv_breakdown_bf16_1024.json 2>/dev/null
