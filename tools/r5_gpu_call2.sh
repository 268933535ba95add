#!/bin/bash
# round 5, GPU call 2: the whole -m gpu suite, the line of record, every named workload, kernel trace + PMC of the headline, the 2-rank dry run
# (exchange overlap model), A/B of the fused kernel at 5 waves per SIMD
mkdir -p gpurun_out/r5b
timeout 1500 python -m pytest tests -m gpu -q --durations=25 > gpurun_out/r5b/pytest_gpu.txt 2>&1; echo "pytest rc=$?"
tail -40 gpurun_out/r5b/pytest_gpu.txt
timeout 600 python bench.py > gpurun_out/r5b/bench_default.json 2> gpurun_out/r5b/bench_default.err; echo "bench rc=$?"
tail -c 1500 gpurun_out/r5b/bench_default.json
python tools/ab.py --rounds 2 --workload reblur_ds_4k base fused_w5 fused_w5_d2 > gpurun_out/r5b/ab_fused_waves.txt 2>&1
tail -5 gpurun_out/r5b/ab_fused_waves.txt
bash tools/bench_workloads.sh r05 > gpurun_out/r5b/workloads.log 2>&1
cat gpurun_out/r5b/workloads.log
NRD_BENCH_DRYRUN_BACKEND=gloo timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/r5b/dry2.json 2> gpurun_out/r5b/dry2.err; echo "dry2 rc=$?"
grep "^{" gpurun_out/r5b/dry2.json | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; print(d['value'], c.get('tiled_bit_identical'), c.get('band_rows'), c.get('halo_exchange_bytes_per_frame_rank0')); print(json.dumps(c.get('exchange_overlap_model'))[:1500])"
bash tools/profile_gpu.sh r05 reblur_ds_4k > gpurun_out/r5b/profile.log 2>&1
tail -12 gpurun_out/r5b/profile.log
cat gpurun_out/profiles/r05_kernel_steady_reblur_ds_4k.csv
