#!/bin/bash
# End-of-round evidence on the final code (run through gpurun; everything lands in gpurun_out/, copy what is judged into profiles/):
#   1. pytest -m gpu (the parity suite, through the C ABI)        2. the driver's bench command
#   3. rocprofv3 --kernel-trace --stats of that command (fp32 512^2) and of the bf16 1024^2 novel-view run
#   4. PMC traffic passes (FETCH_SIZE / WRITE_SIZE, separate runs) for both -> pmc_traffic.json / pmc_traffic_bf16.json
# usage: tools/final_profiles.sh [pytest] [bench] [prof] [pmc]   (default: all)
#   F32_ARGS="--precision winograd" tools/final_profiles.sh prof pmc   -> the fp32-slot profiles of another precision mode of the 512x512 clip
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
R=$PWD; O=gpurun_out; mkdir -p $O
WHAT="${*:-pytest bench prof pmc}"
if [[ " $WHAT " == *" pytest "* ]]; then
  timeout 1500 python -m pytest tests -q -m gpu --durations=8 2>&1 | tail -30 > $O/pytest_gpu.log; echo "pytest exit=${PIPESTATUS[0]}"; tail -4 $O/pytest_gpu.log
fi
if [[ " $WHAT " == *" bench "* ]]; then
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_final.json 2> $O/bench_final.err; echo "bench exit=$?"; head -c 300 $O/bench_final.json; echo
fi
if [[ " $WHAT " == *" prof "* ]]; then
  rm -rf $O/prof $O/prof_bf16
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o lwg -- python $R/bench.py --steps 4 --warmup 2 --cpu-frames 0 --no-conv-events --no-extras --no-self-check ${F32_ARGS:-} > $R/$O/prof_f32.log 2>&1 ); echo "prof f32 exit=$?"
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_bf16 -o lwg -- python $R/bench.py --precision bf16 --size 1024 --workload novel_view --steps 3 --warmup 1 --cpu-frames 0 --no-conv-events --no-extras --no-self-check > $R/$O/prof_bf16.log 2>&1 ); echo "prof bf16 exit=$?"
  for d in prof prof_bf16; do f=$(find $O/$d -name "*kernel_stats*" | head -1); [ -n "$f" ] && cp "$f" $O/${d}_kernel_stats.csv && python tools/prof_summary.py "$f" 6 30 > $O/${d}_summary.txt 2>&1; done
  find $O/prof $O/prof_bf16 -name "*.csv" -size +3M -delete
fi
if [[ " $WHAT " == *" pmc "* ]]; then
  rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_mfma $O/pmc_fetch_bf16 $O/pmc_write_bf16 $O/pmc_mfma_bf16
  BENCH_ARGS="${F32_ARGS:-}" PASSES="A B C" bash tools/pmc_round.sh > $O/final_pmc.log 2>&1
  BENCH_ARGS="--precision bf16 --size 1024 --workload novel_view" TAG=_bf16 KERNEL=lwg_conv_bf16 PASSES="A B C" bash tools/pmc_round.sh > $O/final_pmc_bf16.log 2>&1
  python tools/pmc_summary.py $O/pmc_mfma $O/pmc_mfma.md > /dev/null 2>&1
  python tools/pmc_summary.py $O/pmc_mfma_bf16 $O/pmc_mfma_bf16.md > /dev/null 2>&1
  cat $O/pmc_traffic.json 2>/dev/null | head -c 500; echo; cat $O/pmc_traffic_bf16.json 2>/dev/null | head -c 500; echo
fi
