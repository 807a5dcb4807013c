#!/bin/bash
# Round-6 call 8: baseline of the tree (default bench line with every extra, conv breakdown, per-kernel stats of the fp32 step)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
R=$PWD; O=gpurun_out; mkdir -p $O
timeout 900 python bench.py --conv-breakdown 2>$O/r06_f_bench.err | tail -1 > $O/r06_f_bench_default.json
cp $O/conv_breakdown.json $O/r06_f_conv_breakdown_f32_512.json 2>/dev/null
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06_f_bench_default.json"))
r = d["roofline"]
print(d["value"], "fps", d["ms_per_step"], "ms  frac", r["frac"], "wino", r.get("winograd_kernel_frac"), "up4", r.get("winograd_up4_kernel_frac"), d.get("self_check"))
for k, v in d.items():
    if isinstance(v, dict) and k not in ("roofline", "cpu_baseline", "config"):
        print(k, json.dumps(v)[:400])
PY
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_f32 -o f32 -- python $R/bench.py --steps 2 --warmup 1 --no-extras --cpu-frames 0 --no-self-check > $R/$O/prof_f32.log 2>&1 )
python tools/prof_summary.py $O/prof_f32/f32_kernel_stats.csv 3 24 2>/dev/null | cut -c1-200 | tee $O/r06_f_kernel_stats_f32_512.txt
find $O/prof_f32 -type f -size +3M -delete
