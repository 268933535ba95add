#!/bin/bash
# Runs ON THE GPU BOX: the row-tiling GPU tests (both tilers, the C++ harness's rank modes) and the 2-rank dry runs of the 8K bench. usage: tools/r4_gpu_tiler_check.sh <tag>
TAG=${1:-r04v3}; D=gpurun_out/$TAG; mkdir -p $D
timeout 900 python -m pytest tests/test_tiler_gloo.py tests/test_cpp_harness.py tests/test_band_layout.py -m gpu -q --durations=8 > $D/pytest_tiler_gpu.log 2>&1; echo "pytest rc=$?" | tee $D/status.txt
tail -14 $D/pytest_tiler_gpu.log
for tiler in python native; do
  NRD_BENCH_DRYRUN_BACKEND=gloo timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 4 --warmup 2 --no-cpu-baseline --tiler $tiler > $D/dry2_$tiler.json 2> $D/dry2_$tiler.err
  echo "dryrun2 8K $tiler rc=$?" | tee -a $D/status.txt
  grep "^{" $D/dry2_$tiler.json | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; print(d['value'], c.get('tiled_bit_identical'), c.get('band_rows'), c.get('halo_exchange_bytes_per_frame_rank0'), c.get('predicted_exchange_ms'), str(c.get('native_tiler'))[:300])"
done
