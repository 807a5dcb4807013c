export TMPDIR=/tmp; R=$PWD
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_pers -o pers -- python $R/bench_personalize.py --steps 4 --warmup 2 --no-self-check "$@" > $R/gpurun_out/prof_pers.log 2>&1 )
python tools/prof_summary.py gpurun_out/prof_pers/pers_kernel_stats.csv 6 28
