#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the records of a build - kernel trace + PMC passes of the default bench command (profile_gpu.sh),
# the default bench line, every named workload, the sample-side passes, and the 2-rank dry run of the row-tiled bench on the one GPU.
# Results land under gpurun_out/ (copy the summaries to keep into profiles/). The tag in the file names is edited per build.
mkdir -p gpurun_out/r3v7
bash tools/profile_gpu.sh r03v7 reblur_ds_4k > gpurun_out/r3v7/profile.log 2>&1
timeout 400 python bench.py > gpurun_out/r3v7/bench_default.json 2> gpurun_out/r3v7/bench_default.err
bash tools/bench_workloads.sh r03v7 > gpurun_out/r3v7/workloads.log 2>&1
timeout 200 python bench.py --workload sample_passes_4k --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r3v7/bench_sample_passes.json
NRD_BENCH_DRYRUN_BACKEND=gloo timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/r3v7/dry2.json 2> gpurun_out/r3v7/dry2.err
tail -1 gpurun_out/r3v7/bench_default.json | cut -c1-400; cat gpurun_out/r3v7/workloads.log; ls gpurun_out/profiles | grep r03v7
