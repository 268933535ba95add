#!/bin/bash
# round 5, GPU call 14: ConfidenceBlur with the window decoded once per workgroup into LDS - parity, then the sample-side bench with the old (cb_old) and the new library
mkdir -p gpurun_out/r5n
timeout 300 python -m pytest tests/test_sample_passes.py tests/test_cpp_harness.py -m gpu -q -x -k "confidence or harness_matches" > gpurun_out/r5n/pytest.txt 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r5n/pytest.txt
L=nrd-sample_amd/csrc/libnrdhip.so; cp $L /tmp/new.so
for v in old new old new; do
  if [ $v = old ]; then cp _variants/cb_old.so $L; else cp /tmp/new.so $L; fi
  timeout 200 python bench.py --workload sample_passes_4k --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], {k[8:22]:v for k,v in d['passes_ms'].items()})" | tee -a gpurun_out/r5n/sample_passes_ab.txt
done
cp /tmp/new.so $L
