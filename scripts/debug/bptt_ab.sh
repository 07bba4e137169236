# BPTT kernel A/B: meta-gradient tests + kernel trace of the training step at config-2 shape, T = 100, for a variant library
#   bash scripts/debug/bptt_ab.sh [LIB.so]
[ -n "$1" ] && export L2O_HIP_LIB=$PWD/$1
timeout 900 python -m pytest tests/test_meta_gradient.py -x -q -m gpu 2>&1 | tail -2
bash scripts/trace_train_step.sh gpurun_out/bptt_ab 128 128 100 2>&1 | grep -v "amdgpu.ids\|Warn" | sed -n 4,7p
