#!/bin/bash
# round-4 records, first pass: every named workload on the new default build + kernel trace / PMC of config 4 (RELAX_DIFFUSE_SPECULAR_SH 4K)
bash tools/bench_workloads.sh r04a > gpurun_out/r04a_workloads.log 2>&1
cat gpurun_out/r04a_workloads.log
bash tools/profile_gpu.sh r04a relax_ds_sh_4k 24 > gpurun_out/r04a_profile_relax.log 2>&1
tail -5 gpurun_out/r04a_profile_relax.log
cat gpurun_out/profiles/r04a_kernel_steady_relax_ds_sh_4k.csv
