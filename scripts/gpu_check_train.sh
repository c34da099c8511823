#!/bin/bash
# Training leg on the GPU box: parity tests, train bench (with and without the VQ encode), rocprofv3 kernel stats.
TAG=${1:-r1}
R=$(pwd)
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_train_gpu.py -x -q -rA 2>&1 | tail -40 > gpurun_out/train_tests_$TAG.log
tail -3 gpurun_out/train_tests_$TAG.log
timeout 600 python bench_train.py --steps 3 --warmup 1 > gpurun_out/train_bench_$TAG.log 2>&1
timeout 600 python bench_train.py --steps 3 --warmup 1 --no-vq >> gpurun_out/train_bench_$TAG.log 2>&1
grep '^{' gpurun_out/train_bench_$TAG.log | cut -c1-260
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_train_$TAG -o prof -- python $R/bench_train.py --steps 1 --warmup 1 > $R/gpurun_out/prof_train_$TAG.log 2>&1
cd $R
find gpurun_out/prof_train_$TAG -type f ! -name "*stats*" -size +1M -delete
