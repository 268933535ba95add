#!/bin/bash
# round 5, GPU call 12: occupancy of Blur / PostBlur once more, now that a wave starts faster: PostBlur at 8 waves (1 tap in flight, 62 VGPRs), Blur at 6 waves (3 / 4 taps, 80 / 79 VGPRs)
mkdir -p gpurun_out/r5l
timeout 900 python tools/ab.py --rounds 3 --workload reblur_ds_4k --full-coverage o_base o_post8d1 o_blur6d3 o_blur6d4 > gpurun_out/r5l/ab_headline.txt 2>&1
tail -5 gpurun_out/r5l/ab_headline.txt
