#!/bin/bash
# weight-gradient contraction on the bf16 pipe (k_atb_bx3): microbenchmark, parity tests, training step
TAG=${1:-r03q2}
O=gpurun_out/$TAG; mkdir -p $O
cd "$(dirname "$0")/.."
(timeout 300 scripts/microbench/atb_bx3_bench 1638400 1; timeout 200 scripts/microbench/atb_bx3_bench 327680 1; timeout 300 scripts/microbench/atb_bx3_bench 1638400 2; timeout 60 scripts/microbench/atb_bx3_bench 1000 1) 2>&1 | tee $O/atb_bx3_bench.txt
(timeout 900 python -m pytest tests/test_hip_kernels.py -q -m gpu -k "wgrad or atb" -s 2>&1 | tail -30) | tee $O/pytest_wgrad.log
(timeout 900 python -m pytest tests/test_meta_gradient.py tests/test_rnnprop_gradient.py tests/test_second_derivatives.py tests/test_generic_net.py -q -m gpu 2>&1 | tail -15) | tee $O/pytest_meta.log
(for a in "128 128 20" "128 128 100"; do timeout 300 python scripts/microbench/train_step_timing.py $a; done; L2O_EXACT_GATES=1 timeout 300 python scripts/microbench/train_step_timing.py 128 128 100) 2>&1 | grep -v Warning | tee $O/train_step.txt
