#!/bin/bash
# round 5, GPU call 3: the tiler change (one deferred group per list) on the GPU - tiling tests, the solo latency-injection table - and the bench line again
mkdir -p gpurun_out/r5c
timeout 1200 python -m pytest tests/test_cpp_harness.py tests/test_tiler_gloo.py tests/test_history_rows.py -m gpu -q -s --durations=8 > gpurun_out/r5c/pytest_tiler.txt 2>&1; echo "pytest rc=$?"
grep -v "^EXCHANGE-OVERLAP" gpurun_out/r5c/pytest_tiler.txt | tail -15
cat gpurun_out/exchange_overlap_solo.json
timeout 600 python bench.py > gpurun_out/r5c/bench_default.json 2> gpurun_out/r5c/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r5c/bench_default.json') if l.startswith('{')][-1]); c=d['config']
print(d['value'], d['ms_per_step'], d['passes_ms'])
for k in ('full_coverage','frozen_formulas','hw_transcendentals','without_preroll'):
    print(k, c[k].get('value'), c[k].get('passes_ms'), c[k].get('error'))
PY
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-graph-leg > gpurun_out/r5c/bench_driver_args.json 2>/dev/null
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r5c/bench_driver_args.json') if l.startswith('{')][-1]); c=d['config']
print('driver args:', d['value'], d['ms_per_step'], 'without preroll', c['without_preroll'].get('value'), 'hwt', c['hw_transcendentals'].get('value'), c['hw_transcendentals'].get('passes_ms'))
PY
