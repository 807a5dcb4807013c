export TMPDIR=/tmp; R=$PWD; O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu --durations=6 -k "check_pipeline_full_256 or novel_view_256 or num_source_8_at_512 or only_vis_256 or edge_cases or temporal_mode or lwb_variant_generators or swapper or concat_baselines or source_setup or split_vs_oracle or output_stage or personalize_loop or identity_warp or reference_shape or head_and_layout or batch_slicing" 2>&1 | tail -14 > $O/pytest_gpu_h_rest.log; tail -3 $O/pytest_gpu_h_rest.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_h2.json 2> $O/bench_h2.err; echo "bench exit=$?"; head -c 200 $O/bench_h2.json; echo
rm -rf $O/prof
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o lwg -- python $R/bench.py --steps 4 --warmup 2 --cpu-frames 0 --no-conv-events --no-extras --no-self-check > $R/$O/prof_f32.log 2>&1 ); echo "prof exit=$?"
f=$(find $O/prof -name "*kernel_stats*" | head -1); [ -n "$f" ] && cp "$f" $O/prof_kernel_stats.csv && python tools/prof_summary.py "$f" 6 30 > $O/prof_summary.txt 2>&1
find $O/prof -name "*.csv" -size +3M -delete
head -8 $O/prof_summary.txt
