#!/bin/bash
# round 5, GPU call 7: checkerboard PrepareInputs - one store per output plane (value picked per lane) and uniform-plane loads - against the
# kernel of round 4 (t1_table) and the load-batching attempt (f1_flags); parity of the in-tree library on the PrepareInputs tests
mkdir -p gpurun_out/r5g
timeout 600 python -m pytest tests/test_prepare_inputs.py tests/test_sample_presets.py -m gpu -q -x --durations=3 > gpurun_out/r5g/pytest_prepare.txt 2>&1; echo "pytest rc=$?"
tail -4 gpurun_out/r5g/pytest_prepare.txt
timeout 900 python tools/ab.py --rounds 3 --workload reblur_ds_4k --bench-args=--checkerboard t1_table f1_flags pc_stores pc_both > gpurun_out/r5g/ab_checkerboard.txt 2>&1
tail -5 gpurun_out/r5g/ab_checkerboard.txt
