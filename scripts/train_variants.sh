for v in build/var/lib_*.so; do n=$(basename $v .so); for r in 1 2; do echo -n "$n: "; L2O_HIP_LIB=$PWD/$v python scripts/microbench/train_step_timing.py 128 128 100 2>/dev/null | tail -1; done; done
