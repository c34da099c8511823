#!/bin/bash
# round-3 GPU pass B: accuracy mode (precision 1) parity at tiny + full size, VQ bandwidth line, batch-1 (cfg1 shape) line
TAG=${1:-r3b}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_modules_gpu.py -m gpu -x -q -s -k "accuracy_mode or full_size_logits" 2>&1 | grep -E "parity|passed|failed|Error|error|assert" | tail -60 | tee gpurun_out/${TAG}_precise.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 300 python bench.py --workload vq --steps 10 --warmup 2 > gpurun_out/${TAG}_vq.log 2>&1; grep -h '"metric"' gpurun_out/${TAG}_vq.log | tail -1 > gpurun_out/${TAG}_vq_bench.json; cut -c 1-2500 gpurun_out/${TAG}_vq_bench.json
timeout 600 python bench.py --batch 1 --steps 10 --warmup 2 --no-train-leg --no-cpu-baseline > gpurun_out/${TAG}_b1.log 2>&1; grep -h '"metric"' gpurun_out/${TAG}_b1.log | tail -1 > gpurun_out/${TAG}_b1_bench.json; cut -c 1-400 gpurun_out/${TAG}_b1_bench.json; tail -3 gpurun_out/${TAG}_b1.log | cut -c 1-300
