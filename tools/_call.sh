timeout 900 python -m pytest tests -q -m gpu -k "winograd or check_pipeline_full_512 or whole_clip or generator_golden or benched_shapes_512 or pipeline_tiny or check_conv_transpose or num_source_1_and_8" 2>&1 | tail -4
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_h.json 2> gpurun_out/bench_h.err; echo "bench exit=$?"; head -c 400 gpurun_out/bench_h.json; echo
