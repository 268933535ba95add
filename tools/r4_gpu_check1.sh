#!/bin/bash
# round-4 checkpoint on the GPU box: new harness / tiler tests, the 2-rank dry run of the row-tiled bench (halo volume), the default bench line
mkdir -p gpurun_out/r4j
timeout 900 python -m pytest tests/test_cpp_harness.py tests/test_tiler_gloo.py -m gpu -q --durations=12 > gpurun_out/r4j/pytest_new.log 2>&1; echo "pytest rc=$?"
tail -25 gpurun_out/r4j/pytest_new.log
NRD_BENCH_DRYRUN_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/r4j/dry2.json 2> gpurun_out/r4j/dry2.err; echo "dry2 rc=$?"
grep "^{" gpurun_out/r4j/dry2.json | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; print(d['value'], c.get('tiled_bit_identical'), c.get('native_tiler'), c.get('band_rows'), c.get('halo_exchange_bytes_per_frame_rank0'), c.get('predicted_exchange_ms'))"
tail -3 gpurun_out/r4j/dry2.err
timeout 400 python bench.py > gpurun_out/r4j/bench_default.json 2> gpurun_out/r4j/bench_default.err; echo "bench rc=$?"
tail -c 2500 gpurun_out/r4j/bench_default.json
