#!/bin/bash
# Runs ON THE GPU BOX: the bench line of every named workload (BASELINE.json configs + extras) with the in-tree library, one JSON line each on stdout
#   usage: tools/bench_workloads.sh > gpurun_out/<tag>/bench_workloads.jsonl      (tools/gpu_session.sh <tag> workloads)
ARGS="--no-cpu-baseline --no-frozen-leg --no-young-leg --no-graph-leg"
for wl in reblur_d_1080p reblur_ds_sigma_1440p relax_ds_sh_4k relax_ds_4k reblur_ds_sh_4k reblur_ds_8k; do
  timeout 300 python bench.py --workload $wl $ARGS 2>/dev/null | tail -1
done
timeout 300 python bench.py --workload reblur_ds_4k --checkerboard $ARGS 2>/dev/null | tail -1
timeout 300 python bench.py --workload relax_ds_sh_4k --atrous 8 $ARGS 2>/dev/null | tail -1
timeout 300 python bench.py --workload reblur_ds_4k --roll 90 $ARGS 2>/dev/null | tail -1
