#!/bin/bash
# ONE parametrised GPU-lease script (replaces the per-lease scripts/gpu_r0*.sh of rounds 2-3):
#
#   gpurun --timeout S -- bash scripts/gpu_lease.sh TAG JOB [JOB ...]
#
# Every job writes under gpurun_out/TAG/ (merged back by gpurun; copy what should be judged into profiles/).
# Jobs (run in the order given; a failing job does not stop the next one):
#   tests[:PYTEST_ARGS]      python -m pytest tests -m gpu  (extra args after the colon, '+' separates words, '%' is a
#                            space INSIDE a word: tests:-s+-k+teacher_forced%or%c5_trained)
#   smoke                    __graft_entry__.smoke()
#   bench:CFG[:ARGS]         python bench.py --config CFG (CFG = 2|3|4|5; 2 = default line incl. cpu_baseline)
#   trace:CFG[:ARGS]         rocprofv3 --kernel-trace --stats of the same bench command -> kernel_trace_cCFG.txt
#   counters:CFG[:ARGS]      the PMC passes of scripts/gpu_counters.sh for that bench command -> counters_cCFG.json
#   final:CFG[:ARGS]         bench + trace + counters of ONE config back to back in this lease (same build, same box):
#                            what the roofline block of the round's bench line is recomputed from
#   train:NAME[:SECONDS]     meta-train one committed optimizer (NAME = c2|c3|c4|c5) -> trained/<dir>/
#   env:VAR=VALUE            export VAR=VALUE for the jobs that follow (env:VAR= unsets it), e.g. env:L2O_OPTIONS=one_lds=2 tests:-k+fused
#   bench2ranks              the N > 1 bench path as two gloo ranks on this one device
#   py:SCRIPT[:ARGS]         python SCRIPT ARGS  > SCRIPT-basename.txt
#   sh:COMMAND               bash -c COMMAND      > sh_N.txt   ('+' separates words, as above)
TAG=${1:?usage: gpu_lease.sh TAG JOB [JOB ...]}; shift
cd "$(dirname "$0")/.."
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p $O
words() { echo "${1//+/ }"; }
bench_args() { # CFG -> the bench.py arguments of that BASELINE configuration
  case $1 in
    2) echo "--steps 20 --warmup 5" ;;
    3) echo "--config 3 --steps 5" ;;
    4) echo "--config 4 --steps 10" ;;
    5) echo "--config 5 --steps 5" ;;
    5x8) echo "--config 5 --replicas 8 --steps 5" ;;      # eight replicas per launch, one per XCD (k_mlp_xcd)
    4s8) echo "--config 4 --emulate-world 8 --steps 10" ;; # config 4's shard of 8
    *) echo "--config $1" ;;
  esac
}
prof_args() { # CFG -> the same workload with a SHORT timed region, for the rocprofv3 passes (counters are per launch)
  case $1 in
    2) echo "--steps 20 --warmup 5 --unrolls-per-step 16" ;;
    *) echo "$(bench_args $1) --min-timed-seconds 0.2" ;;
  esac
}
workload_json() { # the `workload` key scripts/counters_to_json.py stores and bench.py matches
  case $1 in
    2) echo '["quadratic", "dm", 128, 128, 100]' ;;
    3) echo '["lasso", "rnnprop", 512, 256, 200, 256]' ;;
    4) echo '["rastrigin", "dm", 100, 1024, 100]' ;;
    5) echo '["mnist", "rnnprop", 15910, 64, 200]' ;;
    5x8) echo '["mnist", "rnnprop", 15910, 64, 200, "replicas", 8]' ;;
    4s8) echo '["rastrigin", "dm", 100, 128, 100]' ;;
  esac
}
kernel_of() { case $1 in 2|4s8) echo 'k_unroll_pair<' ;; 4) echo 'k_unroll_lds<' ;; 3) echo 'k_unroll_cu' ;; 5) echo 'k_mlp_unroll' ;; 5x8) echo 'k_mlp_xcd' ;; esac; }
train_cmd() { # NAME SECONDS -> command line (the ones recorded in tests/golden/trained/README.md)
  local S=$2
  case $1 in
    c2) echo "scripts/train_dm.py --problem quadratic --num_dims 128 --batch_size 128 --num_steps 100 --unroll_length 20 --num_epochs 100000 --evaluation_period 100 --evaluation_epochs 5 --learning_rate 0.001 --seed 1 --max_seconds ${S:-110} --save_path $O/trained/dm_quadratic_d128" ;;
    c4) echo "scripts/train_dm.py --problem rastrigin --num_dims 100 --batch_size 1024 --num_steps 100 --unroll_length 20 --num_epochs 100000 --evaluation_period 50 --evaluation_epochs 3 --learning_rate 0.001 --seed 2 --max_seconds ${S:-150} --save_path $O/trained/dm_rastrigin_d100" ;;
    c3) echo "scripts/train_rnnprop.py --problem lasso --num_dims 512 --num_rows 256 --l 0.1 --batch_size 256 --num_steps 200 --unroll_length 20 --num_epochs 100000 --evaluation_period 20 --evaluation_epochs 2 --learning_rate 0.001 --seed 3 --max_seconds ${S:-150} --save_path $O/trained/rnnprop_lasso_256x512" ;;
    c5) echo "scripts/train_rnnprop.py --problem mnist --synthetic_mnist 4096 --synthetic_seed 5 --synthetic_label_noise 0.1 --batch_size 64 --num_steps 200 --unroll_length 20 --num_epochs 100000 --evaluation_period 20 --evaluation_epochs 3 --learning_rate 0.001 --seed 5 --max_seconds ${S:-150} --save_path $O/trained/rnnprop_mnist_mlp" ;;
  esac
}
n=0
for job in "$@"; do
  kind=${job%%:*}; rest=${job#*:}; [ "$rest" = "$job" ] && rest=""
  a1=${rest%%:*}; a2=${rest#*:}; [ "$a2" = "$rest" ] && a2=""
  echo "=== [$TAG] $job"
  case $kind in
    tests)
      n=$((n + 1))
      IFS='+' read -ra targs <<< "$rest"; targs=("${targs[@]//%/ }")
      timeout 2400 python -m pytest tests -q -m gpu --durations=5 "${targs[@]}" > $O/pytest_full_$n.log 2>&1
      tail -25 $O/pytest_full_$n.log | tee $O/pytest.log ;;
    smoke)
      (timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1) | tee $O/smoke.log ;;
    bench)
      extra=$([ "$a1" = 2 ] || echo --no-cpu-baseline)
      # (a bench job with extra arguments is a different workload: its own file, bench_cN_xK.json)
      n=$((n+1)); out=$O/bench_c$a1.json; [ -n "$a2" ] && out=$O/bench_c${a1}_x$n.json
      timeout 600 python bench.py $(bench_args $a1) $extra $(words "$a2") 2>>$O/bench.err | tee $out | cut -c1-300 ;;
    trace)
      (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_c$a1 -o t -- \
         python $R/bench.py $(prof_args $a1) --no-cpu-baseline --no-also $(words "$a2") > $O/trace_c$a1.json 2> $O/trace_c$a1.err)
      db=$(ls $O/trace_c$a1/*.db $O/trace_c$a1/*/*.db 2>/dev/null | head -1)
      [ -n "$db" ] && python scripts/rocprof_summary.py $db $O/kernel_trace_c$a1.txt | head -8
      rm -rf $O/trace_c$a1 ;;
    counters)
      bash scripts/gpu_counters.sh $O c$a1 "$(kernel_of $a1)" "$(workload_json $a1)" $(prof_args $a1) $(words "$a2") ;;
    final)
      extra=$([ "$a1" = 2 ] || echo --no-cpu-baseline)
      bash scripts/gpu_counters.sh $O c$a1 "$(kernel_of $a1)" "$(workload_json $a1)" $(prof_args $a1) $(words "$a2")
      # the bench line AFTER the counters exist in this lease: its roofline block reads gpurun_out/TAG/counters_cN.json
      L2O_COUNTERS_DIR=$O timeout 600 python bench.py $(bench_args $a1) $extra $(words "$a2") 2>>$O/bench.err | tee $O/bench_c$a1.json | cut -c1-300 ;;
    train)
      mkdir -p $O/trained
      timeout 900 python $(train_cmd $a1 $a2) > $O/train_$a1.log 2>&1
      grep -E "eval_loss|Saving|total time" $O/train_$a1.log | tail -6 ;;
    env)
      if [ -z "${rest#*=}" ]; then unset "${rest%%=*}"; else export "$rest"; fi ;;
    bench2ranks)
      L2O_BENCH_BACKEND=gloo L2O_BENCH_ONE_DEVICE=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
        --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 2 2>$O/bench2.err | tee $O/bench_2ranks_one_device.json | cut -c1-260 ;;
    py)
      (timeout 1200 python $a1 $(words "$a2") 2>&1 | grep -v "amdgpu.ids\|UserWarning") | tee $O/$(basename ${a1%.py}).txt | tail -40 ;;
    sh)
      n=$((n + 1)); (timeout 1200 bash -c "$(words "$rest")" 2>&1) | tee $O/sh_$n.txt | tail -40 ;;
    *) echo "unknown job $job" ;;
  esac
done
