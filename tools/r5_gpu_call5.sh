#!/bin/bash
# round 5, GPU call 5: ClassifyTiles' flags in launch order (FrameConsts::tileFlags) A/B, the restructured checkerboard PrepareInputs A/B, parity
mkdir -p gpurun_out/r5e
timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_prepare_inputs.py tests/test_graph.py -m gpu -q -x -k "not 8k and not 4k" --durations=5 > gpurun_out/r5e/pytest_parity.txt 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r5e/pytest_parity.txt
timeout 900 python tools/ab.py --rounds 3 --workload reblur_ds_4k --full-coverage f0_noflags f1_flags > gpurun_out/r5e/ab_headline.txt 2>&1
tail -3 gpurun_out/r5e/ab_headline.txt
timeout 600 python tools/ab.py --rounds 2 --workload reblur_ds_4k --bench-args "--checkerboard" t1_table f0_noflags f1_flags > gpurun_out/r5e/ab_checkerboard.txt 2>&1
tail -4 gpurun_out/r5e/ab_checkerboard.txt
timeout 600 python tools/ab.py --rounds 2 --workload relax_ds_sh_4k f0_noflags f1_flags > gpurun_out/r5e/ab_relax_sh.txt 2>&1
tail -3 gpurun_out/r5e/ab_relax_sh.txt
timeout 600 python tools/ab.py --rounds 2 --workload reblur_d_1080p f0_noflags f1_flags > gpurun_out/r5e/ab_1080p.txt 2>&1
tail -3 gpurun_out/r5e/ab_1080p.txt
