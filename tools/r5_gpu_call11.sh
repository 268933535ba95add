#!/bin/bash
# round 5, GPU call 11: sequential history footprints in TemporalAccumulation of the SH / RELAX flavours too (n_all) against the REBLUR radiance flavours only (n_base)
mkdir -p gpurun_out/r5k
for wl in relax_ds_sh_4k relax_ds_4k reblur_ds_sh_4k; do
timeout 600 python tools/ab.py --rounds 2 --workload $wl n_base n_all > gpurun_out/r5k/ab_$wl.txt 2>&1
tail -3 gpurun_out/r5k/ab_$wl.txt
done
