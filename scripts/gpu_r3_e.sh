#!/bin/bash
# round-3 GPU pass E: weight-gradient GEMM on token-major operands (gemm_tn.hip) -- harness, tests, training-step A/B
TAG=${1:-r3e}
mkdir -p gpurun_out
timeout 300 ./tools/gemm_bench 5 5 2>&1 | tee gpurun_out/${TAG}_tn_harness.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_train_gpu.py -m gpu -x -q -k "gemm_tn or train" > gpurun_out/${TAG}_tests.log 2>&1
grep -E "passed|failed|Error|error" gpurun_out/${TAG}_tests.log | tail -8
for cfg in "SHOWO_TRAIN_TN=1" "SHOWO_TRAIN_TN=0" "SHOWO_TRAIN_TN=1" "SHOWO_TRAIN_TN=0"; do
  env $cfg timeout 400 python bench.py --workload train --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_train.log 2>&1
  echo "$cfg"; grep -h '"metric"' gpurun_out/${TAG}_train.log | tail -1 | cut -c 1-160
done
