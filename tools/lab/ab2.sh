cd /root/repo
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "conv_backward or generator_training_grads or discriminator_and_trainer_step or personalize_loop or vgg_loss or face_loss or discriminator_variants or graph_vs_eager" > gpurun_out/ab_checks.log 2>&1; echo "checks exit=$?" >> gpurun_out/ab_checks.log
rm -f gpurun_out/ab_pers.log
for rep in 1 2; do
for v in "--no-fused-bias" ""; do
  python bench_personalize.py --steps 30 --warmup 5 $v 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$v]', d.get('ms_per_step'), d.get('roofline',{}).get('frac'))" >> gpurun_out/ab_pers.log
done
done
cat gpurun_out/ab_pers.log; tail -5 gpurun_out/ab_checks.log
