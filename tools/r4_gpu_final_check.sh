#!/bin/bash
# Runs ON THE GPU BOX: a bit-exact kernel change re-verified in one bounded session - A/B against the previous build (_variants/cur.so, _variants/mask.so),
# the parity tests of every denoiser family, the default bench line and the named workloads. usage: tools/r4_gpu_final_check.sh <tag>
TAG=${1:-r04v5}; D=gpurun_out/$TAG; mkdir -p $D
bash tools/ab_r4.sh 2 cur mask > $D/ab.txt 2>&1; cat $D/ab.txt
timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_golden.py tests/test_frozen_flavour.py tests/test_known_answers.py tests/test_prepare_inputs.py tests/test_settings_variants.py -m gpu -q --durations=5 > $D/pytest_gpu_subset.log 2>&1; echo "pytest rc=$?" | tee $D/status.txt
tail -9 $D/pytest_gpu_subset.log
timeout 400 python bench.py > $D/bench_default.json 2> $D/bench_default.err; echo "bench rc=$?" | tee -a $D/status.txt
bash tools/bench_workloads.sh $TAG > $D/workloads.log 2>&1; cat $D/workloads.log
python - $D/bench_default.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c = d["config"]; r = d["roofline"]
print(sys.argv[1], d["value"], d["ms_per_step"], "dominant", r["kernel"], r["frac"], "pipeline", r["pipeline_frac_contract"], "traffic", r["traffic"])
print("  full coverage", c["full_coverage"]["value"], c["full_coverage"]["pipeline_frac_contract"], "| frozen", c["frozen_formulas"]["value"], c["frozen_formulas"].get("distance_from_default"))
print("  passes", d["passes_ms"])
PY
