mkdir -p gpurun_out/o
(timeout 200 python -m pytest tests/test_hip_kernels.py -q -m gpu -x -k "streaming or shared_matrix or c3" 2>&1 | tail -3)
for r in 256 16; do echo rows=$r; timeout 100 python bench.py --steps 5 --warmup 2 --problem lasso --net rnnprop --dims 512 --rows $r --batch 256 --unroll 200 --no-cpu-baseline 2>>gpurun_out/o/bench.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value']/1e9, d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['final_loss_fx_T'])"; done
