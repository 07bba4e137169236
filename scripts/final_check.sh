#!/bin/bash
# round-end style verification: GPU tests, smoke, default bench, the N > 1 code path (2 ranks on one device, gloo)
O=gpurun_out/final; mkdir -p $O
(timeout 300 python -m pytest tests -q -m gpu -x 2>&1 | tail -3) | tee $O/pytest.log
(timeout 60 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1) | tee $O/smoke.log
timeout 200 python bench.py 2>$O/bench.err | tee $O/bench_c2.json | cut -c1-260
L2O_BENCH_BACKEND=gloo L2O_BENCH_ONE_DEVICE=1 timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 2 2>$O/bench2.err | tee $O/bench_2ranks_one_device.json | cut -c1-260
