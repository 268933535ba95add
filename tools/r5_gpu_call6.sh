#!/bin/bash
# round 5, GPU call 6: restructured checkerboard PrepareInputs A/B (t1_table = old kernel), TemporalStabilization at 8 waves, ClassifyTiles run length
mkdir -p gpurun_out/r5f
timeout 600 python tools/ab.py --rounds 2 --workload reblur_ds_4k --bench-args=--checkerboard t1_table f1_flags > gpurun_out/r5f/ab_checkerboard.txt 2>&1
tail -3 gpurun_out/r5f/ab_checkerboard.txt
timeout 900 python tools/ab.py --rounds 3 --workload reblur_ds_4k f1_flags ts8 ct2 ct8 > gpurun_out/r5f/ab_ts_ct.txt 2>&1
tail -5 gpurun_out/r5f/ab_ts_ct.txt
