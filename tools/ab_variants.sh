#!/bin/bash
# A/B timing of library variants on ONE box: alternate the variants, print per-pass times
for round in $(seq 1 ${ROUNDS:-3}); do
  for v in "$@"; do
    cp _variants/$v.so nrd-sample_amd/csrc/libnrdhip.so
    timeout 200 python bench.py --workload ${WL:-reblur_ds_4k} --no-cpu-baseline --no-full-coverage --no-frozen-leg --no-young-leg --no-graph-leg ${BENCH_ARGS:-} 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['passes_ms']
print('$v', round(d['value']), ' '.join('%s=%.4f' % (k.split('::')[1][:6], v) for k, v in p.items()))"
  done
done
