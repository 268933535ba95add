#!/bin/bash
# round 5, GPU call 15: Compose with the sRGB -> linear / unorm8 look-up tables - parity, then the sample-side bench with the old (cb_old) and the new library
mkdir -p gpurun_out/r5o
timeout 300 python -m pytest tests/test_frontend.py tests/test_sample_passes.py -m gpu -q -x > gpurun_out/r5o/pytest.txt 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r5o/pytest.txt
L=nrd-sample_amd/csrc/libnrdhip.so; cp $L /tmp/new.so
for v in old new old new; do
  if [ $v = old ]; then cp _variants/cb_old.so $L; else cp /tmp/new.so $L; fi
  timeout 200 python bench.py --workload sample_passes_4k --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], {k[8:22]:v for k,v in d['passes_ms'].items()})" | tee -a gpurun_out/r5o/sample_passes_ab.txt
done
cp /tmp/new.so $L
