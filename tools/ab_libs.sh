#!/bin/bash
# lab: whole-library A/B under the personalization bench inside ONE gpurun call on ONE box (boxes differ by a few per cent):
#   tools/lab/liblwg_head.so and tools/lab/liblwg_new.so (variant builds, e.g. -DLWG_CONV_SPLIT_MAX_TILES=1024) are swapped under
#   bench_personalize.py, A/B/A/B; a few training checks run first on the library in the tree.  Results: gpurun_out/ab_pers.log
cd /root/repo
mkdir -p gpurun_out
L=ipercore_amd/liblwg_hip.so
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "conv_backward or generator_training_grads or discriminator_and_trainer_step or personalize_loop" > gpurun_out/ab_checks.log 2>&1; echo "checks exit=$?" >> gpurun_out/ab_checks.log
rm -f gpurun_out/ab_pers.log
for rep in 1 2; do
for v in head new; do
  cp tools/lab/liblwg_$v.so $L
  python bench_personalize.py --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d.get('ms_per_step'), d.get('roofline',{}).get('frac'))" >> gpurun_out/ab_pers.log
done
done
cp tools/lab/liblwg_new.so $L
cat gpurun_out/ab_pers.log; grep -n "passed\|failed\|Error" gpurun_out/ab_checks.log | head
