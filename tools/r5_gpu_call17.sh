#!/bin/bash
# round 5, GPU call 17: centre tap texels of Blur / PostBlur as ONE 16-byte buffer load (the compiler split them around the per-pixel sky test) - A/B
mkdir -p gpurun_out/r5q
timeout 700 python tools/ab.py --rounds 3 --workload reblur_ds_4k --full-coverage c0 c1 > gpurun_out/r5q/ab_headline.txt 2>&1
tail -3 gpurun_out/r5q/ab_headline.txt
