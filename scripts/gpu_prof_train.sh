#!/bin/bash
# rocprofv3 kernel table of the training bench (tuner off, like for like with profiles/r3g_train_kernel_stats_tuner_off.csv)
TAG=${1:-r3h}
R=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
SHOWO_GEMM_TUNE=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_train -o prof -- python $R/bench.py --workload train --steps 3 --warmup 1 --no-cpu-baseline --no-events > $R/gpurun_out/prof_${TAG}_train.log 2>&1
cd $R
find gpurun_out/prof_${TAG}_train -type f ! -name "*stats*" -size +2M -delete
head -40 gpurun_out/prof_${TAG}_train/prof_kernel_stats.csv | cut -c1-150
