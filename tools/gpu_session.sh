#!/bin/bash
# One parametrised GPU session (runs ON THE GPU BOX through gpurun): tools/gpu_session.sh <tag> <step> [<step> ...]
# Every step writes under gpurun_out/<tag>/ (merged back into the container; copy what is to be kept into profiles/). Steps:
#   parity                 pytest -m gpu of the parity files (test_parity_gpu, test_steady_state, test_ortho, test_kernels on the device)
#   suite                  the whole pytest -m gpu suite (what the driver runs at round end)
#   pytest:<expr>          pytest -m gpu -k <expr>
#   bench                  python bench.py (the default command)          -> bench_default.json
#   driverbench            python bench.py --steps 20 --warmup 5          -> bench_driver_args.json
#   workloads              tools/bench_workloads.sh                       -> bench_workloads.jsonl
#   samplepasses           bench.py --workload sample_passes_4k           -> bench_sample_passes.json
#   profile:<workload>     tools/profile_gpu.sh <tag> <workload> (kernel trace + PMC passes + HBM traffic) -> gpurun_out/profiles/<tag>_*
#   ab:<v1>,<v2>,...[@workload]   tools/ab.py --rounds 3 --full-coverage over _variants/<v>.so      -> ab_<v1>_..txt
#   abq:<v1>,<v2>,...[@workload]  the same without the no-sky leg (quicker)
#   smoke                  __graft_entry__.smoke()
#   sh:<command>           any shell command (quote it)
set -u
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
for STEP in "$@"; do
    NAME=${STEP%%:*}; ARG=${STEP#*:}; [ "$ARG" = "$STEP" ] && ARG=""
    echo "==== $STEP"
    case $NAME in
    parity) timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_steady_state.py tests/test_ortho.py -m gpu -q -x --durations=5 > $OUT/pytest_parity.txt 2>&1; echo "rc=$?"; tail -3 $OUT/pytest_parity.txt ;;
    suite) timeout 1500 python -m pytest tests -m gpu -q --durations=10 > $OUT/pytest_gpu.txt 2>&1; echo "rc=$?"; tail -5 $OUT/pytest_gpu.txt ;;
    pytest) timeout 900 python -m pytest tests -m gpu -q -x -k "$ARG" > $OUT/pytest_k.txt 2>&1; echo "rc=$?"; tail -5 $OUT/pytest_k.txt ;;
    bench) timeout 400 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "rc=$?"; python tools/print_bench.py $OUT/bench_default.json ;;
    driverbench) timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2> $OUT/bench_driver_args.err; echo "rc=$?"; python tools/print_bench.py $OUT/bench_driver_args.json ;;
    workloads) timeout 900 bash tools/bench_workloads.sh > $OUT/bench_workloads.jsonl 2> $OUT/bench_workloads.err; echo "rc=$?"; python tools/print_bench.py $OUT/bench_workloads.jsonl ;;
    samplepasses) timeout 300 python bench.py --workload sample_passes_4k > $OUT/bench_sample_passes.json 2> $OUT/bench_sample_passes.err; echo "rc=$?"; cat $OUT/bench_sample_passes.json ;;
    profile) timeout 1500 bash tools/profile_gpu.sh $TAG $ARG > $OUT/profile_$ARG.log 2>&1; echo "rc=$?"; cat gpurun_out/profiles/${TAG}_kernel_steady_$ARG.csv 2>/dev/null ;;
    ab|abq)
        WL=reblur_ds_4k; V=$ARG
        case $ARG in *@*) WL=${ARG#*@}; V=${ARG%@*} ;; esac
        FC="--full-coverage"; [ $NAME = abq ] && FC=""
        F=$OUT/ab_$(echo $V | tr ',' '_')_$WL.txt
        timeout 1500 python tools/ab.py --rounds 3 --workload $WL $FC $(echo $V | tr ',' ' ') > $F 2>&1; echo "rc=$?"; sed -n '/^---- medians/,$p' $F ;;
    smoke) timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; echo "rc=$?"; tail -2 $OUT/smoke.txt ;;
    sh) bash -c "$ARG" ;;
    *) echo "unknown step $STEP" ;;
    esac
done
