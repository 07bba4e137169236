#!/bin/bash
# meta-training throughput with the reference's training schedule after the host-side changes of round 3 (same commands as
# tests/golden/trained/README.md, 30 s each; the committed golden weights are NOT replaced)
TAG=${1:-r03x}
OUT=gpurun_out/$TAG
mkdir -p $OUT/trained
cd "$(dirname "$0")/.."
python scripts/train_dm.py --problem quadratic --num_dims 128 --batch_size 128 --num_steps 100 --unroll_length 20 \
  --num_epochs 100000 --evaluation_period 100 --evaluation_epochs 5 --learning_rate 0.001 --seed 1 \
  --max_seconds 30 --save_path $OUT/trained/dm_quadratic_d128 > $OUT/train_dm_quadratic.log 2>&1
grep -E "eval_loss|total time" $OUT/train_dm_quadratic.log | tail -4
python scripts/train_dm.py --problem rastrigin --num_dims 100 --batch_size 1024 --num_steps 100 --unroll_length 20 \
  --num_epochs 100000 --evaluation_period 50 --evaluation_epochs 3 --learning_rate 0.001 --seed 2 \
  --max_seconds 30 --save_path $OUT/trained/dm_rastrigin_d100 > $OUT/train_dm_rastrigin.log 2>&1
grep -E "eval_loss|total time" $OUT/train_dm_rastrigin.log | tail -4
python scripts/train_rnnprop.py --problem lasso --num_dims 512 --num_rows 256 --l 0.1 --batch_size 256 --num_steps 200 \
  --unroll_length 20 --num_epochs 100000 --evaluation_period 20 --evaluation_epochs 2 --learning_rate 0.001 --seed 3 \
  --max_seconds 30 --save_path $OUT/trained/rnnprop_lasso_256x512 > $OUT/train_rnnprop_lasso.log 2>&1
grep -E "eval_loss|total time" $OUT/train_rnnprop_lasso.log | tail -4
rm -rf $OUT/trained
