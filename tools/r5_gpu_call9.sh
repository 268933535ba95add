#!/bin/bash
# round 5, GPU call 9: fused PrePass + TemporalAccumulation with the pixel's motion vector fetched in front of the PrePass taps - A/B
mkdir -p gpurun_out/r5i
timeout 900 python tools/ab.py --rounds 3 --workload reblur_ds_4k --full-coverage mv0 mv1 > gpurun_out/r5i/ab_headline.txt 2>&1
tail -3 gpurun_out/r5i/ab_headline.txt
timeout 600 python tools/ab.py --rounds 2 --workload reblur_d_1080p mv0 mv1 > gpurun_out/r5i/ab_1080p.txt 2>&1
tail -3 gpurun_out/r5i/ab_1080p.txt
timeout 300 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "hip_matches_oracle" > gpurun_out/r5i/pytest_parity.txt 2>&1; echo "pytest rc=$?"
tail -2 gpurun_out/r5i/pytest_parity.txt
