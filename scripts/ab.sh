#!/bin/bash
# ONE parametrised A/B script for the kernel-form comparisons of round 4 (replaces ten one-off ab_*.sh / ablate_lds.sh):
#
#   gpurun -- bash scripts/ab.sh NAME [OUTDIR]          -> OUTDIR/NAME.txt, one line per (library variant, option, workload)
#
# A "library variant" is every build/var/lib_*.so (scripts/build_variants.sh name:-DFLAG ...; lib_a_base.so = the in-tree
# build); options are the binding's environment switches (open_l2o_amd/_abi.py).  NAMEs and what they were used for:
#   large_shard_forms   L2O_ONE_LDS = 0 chunked two-CU | 2 k_unroll_lds, config 4 and config-2 shape x 256..1024
#   rnnprop_large       RNNProp on large shards: chunked two-CU form vs k_unroll_lds
#   lds_variants        build variants of k_unroll_lds (unpinned re-arm, MFMA order, s_setprio): config-2 shape x 256, config 4
#   ablate_lds          timing-only ablations of k_unroll_lds (-DL2O_LDS_ABL_NOBAR / _NOFRAG / _NOGEMV / _ANYNW builds)
#   pk                  packed GEMV FMAs (-DL2O_GEMV_PK=0 variant) over the three fused forms
#   cu_forms            L2O_UNROLL_CU = 2 four-wave | 3 / 4 eight-wave streaming kernel, config 3 (+ DM and D = 256 shapes)
#   cu8_ring            k_unroll_cu8: ring depth (variants -DL2O_CU8_RING=n) x register tiles (L2O_UNROLL_CU = 3 | 4), config 3
#   cu_fourwave         the four-wave streaming kernel (L2O_UNROLL_CU=2) over the library variants, config 3
#   onecu_vs_pair       L2O_NO_PAIR / L2O_ONE_LDS forms at config-2 / config-4 sizes
#   c3_libs             config 3 (default form) over the library variants, e.g. -DL2O_CU8_NT=1 (non-temporal matrix stream)
#   c3dm_libs           the DM nets on config 3's optimizee (Lasso 256 x 512, batch 256) over the library variants
#   c5_hier             l2o_mlp_unroll: hierarchical all-reduce on / off (L2O_NO_MLP_HIER=1) + library variants, config 5
NAME=${1:?usage: ab.sh NAME [OUTDIR]}; O=${2:-gpurun_out/ab}; mkdir -p $O
cd "$(dirname "$0")/.."
# run LABEL <bench.py arguments>: one bench line -> one summary line (options in effect are part of the label)
run() {
  local lbl=$1; shift
  python bench.py --warmup 3 --no-cpu-baseline --no-also "$@" 2>>$O/err.txt | LBL="$lbl" ARGS="$*" python -c "
import json,os,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; T=d['config'].get('T',100) if isinstance(d.get('config'),dict) else 100
print('%-34s %-44s kernel_ms=%.4f  value=%.4g G  fx_T=%r  [%s]' % (os.environ['LBL'], os.environ['ARGS'], r['kernel_ms_avg'], d['value']/1e9, d['final_loss_fx_T'], r['kernel'][:36]))" | tee -a $O/$NAME.txt
}
libs() { ls build/var/lib_*.so 2>/dev/null; }
case $NAME in
  large_shard_forms)
    for rep in 1 2; do
      for f in 0 2; do export L2O_OPTIONS=one_lds=$f
        run "ONE_LDS=$f" --config 4 --steps 6; run "ONE_LDS=$f" --batch 256 --steps 10
        run "ONE_LDS=$f" --batch 512 --steps 6 --unrolls-per-step 8; run "ONE_LDS=$f" --batch 1024 --steps 4 --unrolls-per-step 4
      done
      L2O_OPTIONS=one_lds=0 run "ONE_LDS=0" --batch 128 --steps 10
    done ;;
  rnnprop_large)
    for rep in 1 2; do for f in 0 1; do export L2O_OPTIONS=one_lds=$f
      run "ONE_LDS=$f" --net rnnprop --untrained --batch 256 --steps 10
      run "ONE_LDS=$f" --net rnnprop --untrained --batch 1024 --steps 4 --unrolls-per-step 4
      run "ONE_LDS=$f" --net rnnprop --untrained --problem rastrigin --dims 100 --batch 1024 --steps 4 --unrolls-per-step 4
    done; done ;;
  lds_variants)
    for rep in 1 2; do for v in $(libs); do export L2O_HIP_LIB=$PWD/$v
      run "$(basename $v .so)" --batch 256 --steps 10; run "$(basename $v .so)" --config 4 --steps 6
    done; done ;;
  ablate_lds)
    export L2O_OPTIONS=one_lds=2
    for v in $(libs); do export L2O_HIP_LIB=$PWD/$v; l=$(basename $v .so)
      run "$l" --batch 256 --steps 10
      case $l in *anynw*) for d in 64 80 96 112; do run "$l" --batch 256 --dims $d --steps 10; done ;; esac
    done ;;
  pk)
    for rep in 1 2; do for v in $(libs); do export L2O_HIP_LIB=$PWD/$v; l=$(basename $v .so); unset L2O_OPTIONS
      run "$l" --steps 20; run "$l" --config 4 --steps 6; run "$l" --batch 256 --steps 10
    done; done ;;
  cu_forms)
    for rep in 1 2; do for f in 2 3 4; do L2O_OPTIONS=unroll_cu=$f run "UNROLL_CU=$f" --config 3 --steps 4; done; done
    for f in 2 3; do export L2O_OPTIONS=unroll_cu=$f
      run "UNROLL_CU=$f" --problem lasso --net dm --untrained --dims 512 --rows 256 --batch 256 --unroll 100 --steps 4
      run "UNROLL_CU=$f" --problem lasso --net rnnprop --untrained --dims 256 --rows 128 --batch 256 --unroll 100 --steps 4
    done ;;
  cu8_ring)
    for rep in 1 2; do
      for v in $(libs); do export L2O_HIP_LIB=$PWD/$v
        for f in 3 4; do L2O_OPTIONS=unroll_cu=$f run "$(basename $v .so) UNROLL_CU=$f" --config 3 --steps 4; done
      done
      L2O_OPTIONS=unroll_cu=2 run "four-wave" --config 3 --steps 4
    done ;;
  cu_fourwave)
    export L2O_OPTIONS=unroll_cu=2
    for v in $(libs); do for rep in 1 2; do L2O_HIP_LIB=$PWD/$v run "$(basename $v .so) four-wave" --config 3 --steps 4; done; done ;;
  c3_libs)
    for rep in 1 2; do for v in $(libs); do L2O_HIP_LIB=$PWD/$v run "$(basename $v .so)" --config 3 --steps 4; done; done ;;
  c3dm_libs)
    for rep in 1 2; do for v in $(libs); do
      L2O_HIP_LIB=$PWD/$v run "$(basename $v .so) dm" --problem lasso --net dm --dims 512 --rows 256 --batch 256 --unroll 200 --steps 4 --untrained
      L2O_HIP_LIB=$PWD/$v run "$(basename $v .so) dm_logsign" --problem lasso --net dm_logsign --dims 512 --rows 256 --batch 256 --unroll 200 --steps 4 --untrained
    done; done ;;
  onecu_vs_pair)
    for rep in 1 2; do
      L2O_OPTIONS=one_lds=0 run "c4 pair-chunks" --config 4 --steps 10; run "c4 default" --config 4 --steps 10
      L2O_OPTIONS=one_lds=0 run "B=256 pair-chunks" --batch 256 --steps 10; run "B=256 default" --batch 256 --steps 10
      L2O_OPTIONS=pair=0 run "B=128 no two-CU form" --steps 10; L2O_OPTIONS=one_lds=2 run "B=128 k_unroll_lds" --steps 10
    done ;;
  c5_hier)
    for rep in 1 2; do
      unset L2O_HIP_LIB L2O_OPTIONS; run "in-tree hier=on" --config 5 --steps 5
      L2O_OPTIONS=mlp_hier=0 run "in-tree hier=off (runtime)" --config 5 --steps 5
      for v in $(libs); do L2O_HIP_LIB=$PWD/$v run "$(basename $v .so)" --config 5 --steps 5; done
    done ;;
  *) echo "unknown A/B $NAME"; exit 2 ;;
esac
