#!/bin/bash
# deferred losses + device-side problem sampling: parity of the meta API / training tests, epoch timing
TAG=${1:-r03w}
O=gpurun_out/$TAG; mkdir -p $O
cd "$(dirname "$0")/.."
(timeout 900 python -m pytest tests/test_meta_api.py tests/test_meta_gradient.py tests/test_imitation.py tests/test_generic_net.py tests/test_second_derivatives.py tests/test_rnnprop_gradient.py tests/test_mlp_unroll.py -q -m gpu 2>&1 | tail -12) | tee $O/pytest_meta.log
(python scripts/microbench/train_epoch_timing.py; python scripts/microbench/train_epoch_timing.py 128 128 100 1 40) 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $O/train_epoch.txt
