#!/bin/bash
# End-of-round evidence: rocprofv3 kernel stats of the default bench command, the PMC traffic passes, the per-kernel table,
# and bench lines at 256 / 1024 / split / bf16.  Everything lands in gpurun_out/; copy what is judged into profiles/.
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
rm -rf gpurun_out/prof gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_mfma
tools/gpu_round.sh prof > gpurun_out/final_prof.log 2>&1
PASSES="A B C" tools/pmc_round.sh > gpurun_out/final_pmc.log 2>&1
f=$(find gpurun_out/prof -name "*kernel_stats*" | head -1)
python tools/kernel_table.py "$f" gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/kernel_table.md > /dev/null 2>&1
python tools/prof_summary.py "$f" 6 30 > gpurun_out/prof_summary.txt 2>&1
python tools/pmc_summary.py gpurun_out/pmc_mfma gpurun_out/pmc_mfma.md > /dev/null 2>&1
for cfg in "256 --size 256" "1024 --size 1024" "512_split --precision split" "512_bf16 --precision bf16" "1024_bf16_novelview --size 1024 --precision bf16 --workload novel_view"; do
  set -- $cfg; name=$1; shift
  timeout 400 python bench.py "$@" --cpu-frames 0 --output-frames 0 --no-split-extra > gpurun_out/bench_$name.json 2> gpurun_out/bench_$name.err
  echo "$name: $(tail -1 gpurun_out/bench_$name.json | cut -c1-160)"
done
head -12 gpurun_out/prof_summary.txt; cat gpurun_out/pmc_traffic.json 2>/dev/null | head -c 600
