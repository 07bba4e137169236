#!/bin/bash
# meta-step enqueued behind the unroll (guarded Adam): parity of the training path + host phases
TAG=${1:-r03v}
O=gpurun_out/$TAG; mkdir -p $O
cd "$(dirname "$0")/.."
(timeout 900 python -m pytest tests/test_meta_gradient.py tests/test_rnnprop_gradient.py tests/test_second_derivatives.py tests/test_meta_api.py tests/test_generic_net.py tests/test_imitation.py tests/test_hip_kernels.py -q -m gpu -k "not long_horizon" 2>&1 | tail -8) | tee $O/pytest_train.log
(python scripts/microbench/train_step_host_phases.py; python scripts/microbench/train_step_host_phases.py 128 128 100) 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $O/host_phases.txt
