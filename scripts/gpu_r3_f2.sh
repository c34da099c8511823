#!/bin/bash
# round-3 GPU pass F2: fused bias-gradient column sums (ln_bwd / dgelu), parallel gn_finalize -- full suite + training step
TAG=${1:-r3f2}
mkdir -p gpurun_out
bash scripts/gpu_tests.sh $TAG
for cfg in "SHOWO_TRAIN_TN=1" "SHOWO_TRAIN_TN=1"; do
  env $cfg timeout 400 python bench.py --workload train --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_train.log 2>&1
  echo "$cfg"; grep -h '"metric"' gpurun_out/${TAG}_train.log | tail -1 | cut -c 1-160
done
