#!/bin/bash
# round-2 GPU pass 3H: training forward through the fused projection launch (save-for-backward form): kernel test, training tests, step-time A/B
TAG=${1:-r3h}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "qkv" 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
timeout 900 python -m pytest tests/test_train_gpu.py tests/test_batch_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
for v in 1 0 1 0; do
SHOWO_TRAIN_FUSED_PROJ=$v timeout 300 python bench.py --workload train --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/bench_train_${TAG}_$v.log 2>&1
echo "fused_proj=$v"; grep -h '"metric"' gpurun_out/bench_train_${TAG}_$v.log | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(round(d['value'],2), 'ms/step', round(d['roofline']['achieved']), 'TF/s gemm', d['config'].get('losses_last_step'))
"
done
