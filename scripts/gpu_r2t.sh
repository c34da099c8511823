#!/bin/bash
# round-2 GPU pass T: full GPU suite (complete log), rocprof kernel stats of the training step and of cfg3
TAG=${1:-r2t}
R=$(pwd)
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/tests_$TAG.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/tests_$TAG.log | tail -3
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_train -o prof -- python $R/bench.py --workload train --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_${TAG}_train.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_t2i512 -o prof -- python $R/bench.py --workload t2i512 --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_${TAG}_t2i512.log 2>&1
cd $R
find gpurun_out/prof_${TAG}_train gpurun_out/prof_${TAG}_t2i512 -type f ! -name "*stats*" -size +2M -delete
head -30 gpurun_out/prof_${TAG}_train/prof_kernel_stats.csv | cut -c1-200
