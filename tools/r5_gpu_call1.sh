#!/bin/bash
# round 5, GPU call 1: A/B of the hwt flavour and the fused-kernel branch variants, then the new steady-state / hwt tolerance tests
mkdir -p gpurun_out/r5a
python tools/ab.py --rounds 3 --workload reblur_ds_4k base hwt prepass_zero_rejected prepass_weight_unconditional ta_blends_unconditional > gpurun_out/r5a/ab_headline.txt 2>&1
tail -8 gpurun_out/r5a/ab_headline.txt
python tools/ab.py --rounds 2 --workload reblur_ds_4k --full-coverage base hwt > gpurun_out/r5a/ab_hwt_full_coverage.txt 2>&1
tail -4 gpurun_out/r5a/ab_hwt_full_coverage.txt
python tools/ab.py --rounds 2 --workload relax_ds_sh_4k base hwt > gpurun_out/r5a/ab_hwt_relax_sh.txt 2>&1
tail -4 gpurun_out/r5a/ab_hwt_relax_sh.txt
timeout 900 python -m pytest tests/test_steady_state.py tests/test_hw_transcendentals.py -m gpu -q -s --durations=8 > gpurun_out/r5a/pytest_new.log 2>&1; echo "pytest rc=$?"
grep -v "^HWT-DISTANCE" gpurun_out/r5a/pytest_new.log | tail -30
