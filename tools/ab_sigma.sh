#!/bin/bash
# Runs ON THE GPU BOX: A/B of prebuilt library variants (_variants/*.so) on BASELINE config 3, printing the SIGMA passes only
# (tools/ab_variants.sh abbreviates pass names and REBLUR's and SIGMA's collide there).
for round in 1 2; do
  for v in v0 v2 v3; do
    cp _variants/$v.so nrd-sample_amd/csrc/libnrdhip.so
    timeout 200 python bench.py --workload reblur_ds_sigma_1440p --no-cpu-baseline --no-full-coverage --no-frozen-leg --no-young-leg --no-graph-leg 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['passes_ms']
print('$v', round(d['value']), ' '.join('%s=%.4f' % (k.replace('SIGMA::','S.')[:12], v) for k, v in p.items() if k.startswith('SIGMA')))"
  done
done
