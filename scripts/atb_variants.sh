for v in build/var/lib_*.so; do n=$(basename $v .so); echo "== $n"; L2O_HIP_LIB=$PWD/$v python scripts/microbench/atb_bench.py 2>/dev/null | grep "T=100" | cut -c1-150; done
