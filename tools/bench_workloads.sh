#!/bin/bash
# Runs ON THE GPU BOX: the bench line of every named workload (BASELINE.json configs + extras) with the in-tree library
#   usage: tools/bench_workloads.sh <tag>   -> gpurun_out/<tag>_bench_workloads.jsonl
TAG=${1:-rXX}; out=gpurun_out/${TAG}_bench_workloads.jsonl; : > $out
for wl in reblur_d_1080p reblur_ds_sigma_1440p relax_ds_sh_4k relax_ds_4k reblur_ds_sh_4k reblur_ds_8k; do
  timeout 300 python bench.py --workload $wl --no-cpu-baseline --no-frozen-leg --no-young-leg --no-graph-leg 2>/dev/null | tail -1 >> $out
done
timeout 300 python bench.py --workload reblur_ds_4k --checkerboard --no-cpu-baseline --no-frozen-leg --no-young-leg --no-graph-leg 2>/dev/null | tail -1 >> $out
timeout 300 python bench.py --workload relax_ds_sh_4k --atrous 8 --no-cpu-baseline --no-frozen-leg --no-young-leg --no-graph-leg 2>/dev/null | tail -1 >> $out
timeout 300 python bench.py --workload reblur_ds_4k --roll 90 --no-cpu-baseline --no-frozen-leg --no-young-leg --no-graph-leg 2>/dev/null | tail -1 >> $out
python - $out <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l); r = d.get("roofline", {})
    print(d["config"]["workload"][:70], d["value"], d["ms_per_step"], r.get("pipeline_frac"), r.get("pipeline_frac_contract"))
PY
