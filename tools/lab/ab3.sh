cd /root/repo
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "conv_variants or conv_backward or generator_training_grads or discriminator_and_trainer_step or personalize_loop or graph_vs_eager or rccl_world1 or lwb_variant" > gpurun_out/ab_checks.log 2>&1; echo "checks exit=$?" >> gpurun_out/ab_checks.log
rm -f gpurun_out/ab_pers.log
for rep in 1 2; do
for v in "--no-relu-mask" ""; do
  python bench_personalize.py --steps 30 --warmup 5 $v 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$v]', d.get('ms_per_step'), d.get('roofline',{}).get('frac'))" >> gpurun_out/ab_pers.log
done
done
cat gpurun_out/ab_pers.log; tail -5 gpurun_out/ab_checks.log
