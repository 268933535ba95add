#!/bin/bash
# round 5, GPU call 10: fused PrePass + TemporalAccumulation with the two history footprints fetched one after the other (32 registers of raw
# texels less): at 4 waves (what the second round trip costs) and at 5 waves per SIMD with 2 PrePass taps in flight (93 VGPRs, no scratch)
mkdir -p gpurun_out/r5j
timeout 900 python tools/ab.py --rounds 3 --workload reblur_ds_4k --full-coverage sq_base sq_seq4 sq_seq4d2 sq_seq5d2 > gpurun_out/r5j/ab_headline.txt 2>&1
tail -5 gpurun_out/r5j/ab_headline.txt
timeout 600 python tools/ab.py --rounds 2 --workload reblur_d_1080p sq_base sq_seq5d2 > gpurun_out/r5j/ab_1080p.txt 2>&1
tail -3 gpurun_out/r5j/ab_1080p.txt
