#!/bin/bash
# round 5, GPU call 16: per-frame roughness table (RoughTerms) in the set-up of the spatial passes - parity, A/B against computing the terms per pixel
mkdir -p gpurun_out/r5p
timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_graph.py -m gpu -q -x -k "not 8k and not 4k and not 1440p" > gpurun_out/r5p/pytest_parity.txt 2>&1; echo "pytest rc=$?"
tail -2 gpurun_out/r5p/pytest_parity.txt
timeout 900 python tools/ab.py --rounds 3 --workload reblur_ds_4k --full-coverage rl0 rl1 > gpurun_out/r5p/ab_headline.txt 2>&1
tail -3 gpurun_out/r5p/ab_headline.txt
timeout 600 python tools/ab.py --rounds 2 --workload relax_ds_sh_4k rl0 rl1 > gpurun_out/r5p/ab_relax_sh.txt 2>&1
tail -3 gpurun_out/r5p/ab_relax_sh.txt
timeout 600 python tools/ab.py --rounds 2 --workload reblur_ds_sigma_1440p rl0 rl1 > gpurun_out/r5p/ab_config3.txt 2>&1
tail -3 gpurun_out/r5p/ab_config3.txt
