#!/bin/bash
# Round 3, lease 1: meta-train the three optimizers of BASELINE configs 2 / 4 / 3 with the repo's own drivers
# (bounded wall time each), then probe parity of the TRAINED optimizers at full size (scripts/trained_parity_probe.py).
# The .l2l files land in gpurun_out/$TAG/trained/<name>/ ; the ones kept are copied to tests/golden/trained/.
TAG=${1:-r03a}
OUT=gpurun_out/$TAG
mkdir -p $OUT/trained
cd "$(dirname "$0")/.."
T_DM=${T_DM:-110}
T_RA=${T_RA:-150}
T_RP=${T_RP:-150}
set -x
python scripts/train_dm.py --problem quadratic --num_dims 128 --batch_size 128 --num_steps 100 --unroll_length 20 \
  --num_epochs 100000 --evaluation_period 100 --evaluation_epochs 5 --learning_rate 0.001 --seed 1 \
  --max_seconds $T_DM --save_path $OUT/trained/dm_quadratic_d128 > $OUT/train_dm_quadratic.log 2>&1
grep -E "eval_loss|Saving|total time" $OUT/train_dm_quadratic.log | tail -8
python scripts/train_dm.py --problem rastrigin --num_dims 100 --batch_size 1024 --num_steps 100 --unroll_length 20 \
  --num_epochs 100000 --evaluation_period 50 --evaluation_epochs 3 --learning_rate 0.001 --seed 2 \
  --max_seconds $T_RA --save_path $OUT/trained/dm_rastrigin_d100 > $OUT/train_dm_rastrigin.log 2>&1
grep -E "eval_loss|Saving|total time" $OUT/train_dm_rastrigin.log | tail -8
python scripts/train_rnnprop.py --problem lasso --num_dims 512 --num_rows 256 --l 0.1 --batch_size 256 --num_steps 200 \
  --unroll_length 20 --num_epochs 100000 --evaluation_period 20 --evaluation_epochs 2 --learning_rate 0.001 --seed 3 \
  --max_seconds $T_RP --save_path $OUT/trained/rnnprop_lasso_256x512 > $OUT/train_rnnprop_lasso.log 2>&1
grep -E "eval_loss|Saving|total time" $OUT/train_rnnprop_lasso.log | tail -8
ls -la $OUT/trained/*
timeout 900 python scripts/trained_parity_probe.py --weights $OUT/trained --out $OUT/probe.json > $OUT/probe.log 2>&1
tail -40 $OUT/probe.log
