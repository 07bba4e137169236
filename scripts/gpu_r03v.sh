#!/bin/bash
# training path: parity + host phases
TAG=${1:-r03v}
shift
O=gpurun_out/$TAG; mkdir -p $O
cd "$(dirname "$0")/.."
(timeout 900 python -m pytest tests/test_meta_gradient.py tests/test_rnnprop_gradient.py tests/test_second_derivatives.py tests/test_meta_api.py tests/test_generic_net.py tests/test_imitation.py tests/test_mlp_unroll.py -q -m gpu 2>&1 | tail -12) | tee $O/pytest_train.log
(python scripts/microbench/train_step_host_phases.py 1024 100 20 rastrigin; python scripts/microbench/train_step_host_phases.py 128 128 20) 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $O/host_phases.txt
