#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the issue-rate evidence of the headline kernels. Separate rocprofv3 --pmc
# passes (kernel trace only, never combined with sys / hip / hsa traces) over a short run of the default bench workload, summarised by
# tools/valu_issue_report.py into gpurun_out/<tag>_valu_issue.txt (copy to profiles/).
#   usage: tools/pmc_valu_issue.sh <tag> [workload] [library under nrd-sample_amd/csrc/, default libnrdhip.so as it stands]
set -u
TAG=${1:-r04}; WL=${2:-reblur_ds_4k}
ROOT=$(pwd); export TMPDIR=/tmp
CMD="python $ROOT/bench.py --workload $WL --steps 16 --warmup 32 --no-cpu-baseline --no-full-coverage --no-frozen-leg --no-young-leg --no-graph-leg ${BENCH_ARGS:-}"
i=0
for set in \
  "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
  "SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY" \
  "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  rm -rf /tmp/vi_${TAG}_$i
  (cd /tmp && timeout 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/vi_${TAG}_$i -o p -- $CMD > $ROOT/gpurun_out/vi_${TAG}_$i.log 2>&1)
  echo "pmc set $i rc=$?"
done
python $ROOT/tools/valu_issue_report.py /tmp/vi_${TAG}_ 3 16 > $ROOT/gpurun_out/${TAG}_valu_issue.txt
cat $ROOT/gpurun_out/${TAG}_valu_issue.txt
