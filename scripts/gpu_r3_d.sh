#!/bin/bash
# round-3 GPU pass D: GroupNorm statistics from the conv epilogue, single-pass cross-entropy, ln_bwd / embed_rank rewrites,
# raw-only projection launch + qk_prep (SHOWO_TRAIN_QKPREP) -- tests, then training-step A/B
TAG=${1:-r3d}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee gpurun_out/${TAG}_gpu_tests.log
for cfg in "SHOWO_TRAIN_QKPREP=0" "SHOWO_TRAIN_QKPREP=1" "SHOWO_CONV_GN_FUSE=0" "SHOWO_TRAIN_QKPREP=0" "SHOWO_TRAIN_QKPREP=1"; do
  env $cfg timeout 400 python bench.py --workload train --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_train.log 2>&1
  echo "$cfg"; grep -h '"metric"' gpurun_out/${TAG}_train.log | tail -1 | cut -c 1-200
done
SHOWO_TRAIN_QKPREP=1 timeout 600 python -m pytest tests/test_train_gpu.py -m gpu -x -q 2>&1 | tail -3
