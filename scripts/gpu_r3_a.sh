#!/bin/bash
# round-3 GPU pass A: new training-parity tests (production T >= 256 branch), graph-cache test, full suite, headline bench
TAG=${1:-r3a}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_train_gpu.py -m gpu -x -q -s -k "small_training or full_width or out_of_range" 2>&1 | grep -E "parity|passed|failed|Error|error|assert" | tail -80 | tee gpurun_out/${TAG}_train_parity.log
timeout 600 python -m pytest tests/test_modules_gpu.py -m gpu -x -q -k "graph" 2>&1 | tail -5
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/${TAG}_gpu_tests.log
timeout 600 python bench.py --steps 10 --warmup 2 > gpurun_out/${TAG}_bench.log 2>&1; grep -h '"metric"' gpurun_out/${TAG}_bench.log | tail -1 > gpurun_out/${TAG}_bench.json; cut -c 1-1500 gpurun_out/${TAG}_bench.json
