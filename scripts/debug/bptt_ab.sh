# BPTT kernel A/B: meta-gradient tests + kernel trace of the training step at config-2 shape, T = 100
timeout 900 python -m pytest tests/test_meta_gradient.py -x -q -m gpu 2>&1 | tail -2
for kind in quadratic; do bash scripts/trace_train_step.sh gpurun_out/bptt_ab 128 128 100 2>&1 | grep -v "amdgpu.ids\|Warn" | sed -n 4,9p; done
timeout 300 python scripts/microbench/train_step_timing.py 128 128 100 2>&1 | tail -1
