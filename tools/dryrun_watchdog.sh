#!/bin/bash
# Runs ON THE GPU BOX: the N > 1 bench path with 2 ranks on the one GPU (gloo moves the rows; numbers meaningless) - once as the driver
# would run it, once with a 2-second --extras-deadline so that the watchdog around the native-tiler leg / bit-identity check fires.
mkdir -p gpurun_out/r3v8
NRD_BENCH_DRYRUN_BACKEND=gloo timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/r3v8/dry2.json 2> gpurun_out/r3v8/dry2.err; echo "rc=$?"
grep "^{" gpurun_out/r3v8/dry2.json | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; print(d['value'], c.get('tiled_bit_identical'), c.get('native_tiler'), c.get('band_rows'))"
NRD_BENCH_DRYRUN_BACKEND=gloo timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 2 --steps 4 --warmup 2 --no-cpu-baseline --extras-deadline 2 > gpurun_out/r3v8/dry2_watchdog.json 2> gpurun_out/r3v8/dry2_watchdog.err; echo "rc=$?"
grep "^{" gpurun_out/r3v8/dry2_watchdog.json | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; print(d['value'], c.get('tiled_bit_identical'), c.get('tiled_bit_identical_detail'), c.get('native_tiler'))"
tail -3 gpurun_out/r3v8/dry2_watchdog.err
