#!/bin/bash
# round-2 GPU pass V: clean kernel breakdown of the training step (GEMM tuner off so that its timing launches do not pollute the stats)
TAG=${1:-r2v}
R=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
SHOWO_GEMM_TUNE=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_train -o prof -- python $R/bench.py --workload train --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_${TAG}_train.log 2>&1
cd $R
find gpurun_out/prof_${TAG}_train -type f ! -name "*stats*" -size +2M -delete
tail -3 gpurun_out/prof_${TAG}_train.log | cut -c1-400
