#!/bin/bash
# round 5, GPU call 13: graph replay with the launch-order tile flags made before the capture (tests + the bench's graph leg); fused kernel with 1 PrePass tap in flight
mkdir -p gpurun_out/r5m
timeout 600 python -m pytest tests/test_graph.py -m gpu -q -x > gpurun_out/r5m/pytest_graph.txt 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r5m/pytest_graph.txt
timeout 300 python bench.py --no-cpu-baseline --no-frozen-leg --no-young-leg --no-full-coverage > gpurun_out/r5m/bench_graph.json 2>/dev/null
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r5m/bench_graph.json') if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], json.dumps(d['config'].get('graph_replay'))[:600])
PY
timeout 600 python tools/ab.py --rounds 3 --workload reblur_ds_4k g_base g_d1 > gpurun_out/r5m/ab_d1.txt 2>&1
tail -3 gpurun_out/r5m/ab_d1.txt
