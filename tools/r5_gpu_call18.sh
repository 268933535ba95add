#!/bin/bash
# round 5, GPU call 18: records of the final library (r05v6): parity subset, default bench line, driver-arguments line, kernel trace + PMC passes
T=r05v6; mkdir -p gpurun_out/$T
timeout 400 python -m pytest tests/test_parity_gpu.py tests/test_steady_state.py -m gpu -q -x -k "not 8k and not relax_sh_720p and not config3_720p" --durations=3 > gpurun_out/$T/pytest_parity.txt 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/$T/pytest_parity.txt
timeout 300 python bench.py > gpurun_out/$T/bench_default.json 2> gpurun_out/$T/bench_default.err; echo "bench rc=$?"
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-graph-leg > gpurun_out/$T/bench_driver_args.json 2>/dev/null; echo "bench(driver args) rc=$?"
bash tools/profile_gpu.sh $T reblur_ds_4k > gpurun_out/$T/profile.log 2>&1
cat gpurun_out/profiles/${T}_kernel_steady_reblur_ds_4k.csv
python tools/r5_print_bench.py gpurun_out/$T/bench_default.json gpurun_out/$T/bench_driver_args.json
