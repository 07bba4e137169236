#!/bin/bash
# recording forward with the history store behind the exchange: parity of the recorded history / meta-gradients + step time
TAG=${1:-r03s2}
O=gpurun_out/$TAG; mkdir -p $O
cd "$(dirname "$0")/.."
(timeout 900 python -m pytest tests/test_meta_gradient.py tests/test_rnnprop_gradient.py tests/test_second_derivatives.py tests/test_meta_api.py tests/test_hip_kernels.py -q -m gpu -k "not long_horizon" 2>&1 | tail -8) | tee $O/pytest_hist.log
(for a in "128 128 20" "128 128 100" "128 128 100"; do timeout 300 python scripts/microbench/train_step_timing.py $a; done) 2>&1 | grep "train step" | tee $O/train_step.txt
