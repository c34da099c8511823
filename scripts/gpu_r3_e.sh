#!/bin/bash
# round-3 GPU pass E: cost of the two halves of the fused [Wqkv ; W1] epilogue (harness) + the tests that pin its results
TAG=${1:-r3e}
mkdir -p gpurun_out
GEMM_BENCH_EPI_SPLIT=1 GEMM_BENCH_VARS=3192,3160,4192 GEMM_BENCH_GNS=4 timeout 600 ./tools/gemm_bench 5:4:4:3192 2 2>&1 | grep -E "^qkv\|fc1 M=" | tee gpurun_out/${TAG}_harness.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "qkv or qk_prep" 2>&1 | grep -E "passed|failed" | tail -2
timeout 900 python -m pytest tests/test_modules_gpu.py tests/test_train_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3
for i in 1 2; do
timeout 300 python bench.py --steps 6 --warmup 2 --no-train-leg --no-cpu-baseline > gpurun_out/${TAG}_bench.log 2>&1
grep -h '"metric"' gpurun_out/${TAG}_bench.log | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d['roofline']; print(round(d['value'], 2), 'img/s  gemm', round(r['achieved']), 'TF/s  attention', round(r['attention']['achieved'], 1), 'TF/s  vq', round(r['vq_conv']['achieved'],1))
"
done
