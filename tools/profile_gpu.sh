#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): kernel-trace stats + PMC passes of the bench command, summarised into
# gpurun_out/profiles/<tag>_*.csv (copy the ones to keep into profiles/). Counter passes are separate runs and never
# combined with sys/hip/hsa traces (MI355X guide, "rocprofv3 PMC slots").
#   usage: tools/profile_gpu.sh <tag> <workload> [steps]
set -u
TAG=${1:-r01}; WL=${2:-reblur_ds_4k}; STEPS=${3:-48}; WARM=${4:-32}  # defaults = the default bench.py command
ROOT=$(pwd); OUT=$ROOT/gpurun_out/profiles; mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $ROOT/bench.py --workload $WL --steps $STEPS --warmup $WARM --no-cpu-baseline --no-full-coverage --no-frozen-leg --no-young-leg --no-graph-leg"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_$TAG -o k -- $CMD > $ROOT/gpurun_out/prof_$TAG.log 2>&1
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES" "TCC_HIT_sum TCC_MISS_sum"; do
    name=$(echo $pass | cut -d' ' -f1)
    timeout 600 rocprofv3 --pmc $pass --output-format csv -d $ROOT/gpurun_out/pmc_${TAG}_$name -o p -- $CMD > $ROOT/gpurun_out/pmc_${TAG}_$name.log 2>&1
done
cd $ROOT
python tools/summarize_profiles.py $TAG $WL $STEPS
