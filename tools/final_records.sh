#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the records of a build - the GPU test suite with durations, kernel trace + PMC passes of the default
# bench command (profile_gpu.sh), the issue-rate counters (pmc_valu_issue.sh), the default bench line (as built and with the driver's
# arguments), every named workload, the sample-side passes, and the 2-rank dry runs of the row-tiled bench on the one GPU (both tilers).
# Results land under gpurun_out/ (copy the summaries to keep into profiles/).   usage: tools/final_records.sh <tag>
TAG=${1:-r04v1}
D=gpurun_out/$TAG; mkdir -p $D
timeout 1300 python -m pytest tests -m gpu -q --durations=15 > $D/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee $D/status.txt
tail -22 $D/pytest_gpu.log
timeout 400 python bench.py > $D/bench_default.json 2> $D/bench_default.err; echo "bench rc=$?" | tee -a $D/status.txt
timeout 400 python bench.py --steps 20 --warmup 5 > $D/bench_driver_args.json 2> $D/bench_driver_args.err; echo "bench(driver args) rc=$?" | tee -a $D/status.txt
bash tools/profile_gpu.sh $TAG reblur_ds_4k > $D/profile.log 2>&1
bash tools/pmc_valu_issue.sh $TAG reblur_ds_4k > $D/valu_issue.log 2>&1
bash tools/bench_workloads.sh $TAG > $D/workloads.log 2>&1
timeout 200 python bench.py --workload sample_passes_4k --no-cpu-baseline 2>/dev/null | tail -1 > $D/bench_sample_passes.json
for tiler in python native; do
  NRD_BENCH_DRYRUN_BACKEND=gloo timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 4 --warmup 2 --no-cpu-baseline --tiler $tiler > $D/dry2_$tiler.json 2> $D/dry2_$tiler.err
  echo "dryrun2 8K $tiler rc=$?" | tee -a $D/status.txt
done
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $D/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $D/status.txt
for f in bench_default bench_driver_args; do python - $D/$f.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c = d["config"]; r = d["roofline"]
print(sys.argv[1], d["value"], d["ms_per_step"], "dominant", r["kernel"], r["frac"], "pipeline", r["pipeline_frac_contract"], "traffic", r["traffic"])
print("  full coverage", c["full_coverage"]["value"], c["full_coverage"]["pipeline_frac_contract"], "| frozen", c["frozen_formulas"]["value"], c["frozen_formulas"].get("distance_from_default"))
print("  passes", d["passes_ms"])
PY
done
cat $D/workloads.log; cat $D/bench_sample_passes.json | cut -c1-600
for tiler in python native; do grep "^{" $D/dry2_$tiler.json | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; print(d['value'], c.get('tiled_bit_identical'), c.get('band_rows'), c.get('halo_exchange_bytes_per_frame_rank0'), c.get('predicted_exchange_ms'), str(c.get('native_tiler'))[:200])"; done
ls gpurun_out/profiles | grep $TAG; cat gpurun_out/profiles/${TAG}_kernel_steady_reblur_ds_4k.csv
