#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O/suite
timeout 1200 python -m pytest tests -q -m gpu -k "check_face_loss or check_generator_training_grads_512_full or check_discriminator_and_trainer_step or check_graph_vs_eager or check_personalize_loop or check_vgg_loss" 2>&1 | tail -25 > $O/suite/pytest_c.log; echo "pytest exit=${PIPESTATUS[0]}"; tail -12 $O/suite/pytest_c.log
timeout 600 python bench_personalize.py --steps 10 --warmup 4 --use-vgg --use-face > $O/pers_vgg_face.json 2> $O/pers_vgg_face.err; echo "pers vgg face exit=$?"; head -c 1500 $O/pers_vgg_face.json; echo; tail -3 $O/pers_vgg_face.err
timeout 600 python bench_personalize.py --steps 10 --warmup 4 > $O/pers.json 2> $O/pers.err; echo "pers exit=$?"; head -c 700 $O/pers.json; echo
