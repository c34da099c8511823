#!/bin/bash
# round-4 pass H: stream-K form of the production GEMM (rule-selected): its tests + the whole GEMM test file (the kernel was restructured into
# segments), harness timings against the one-tile-per-block variants, t2i / training / t2i512 A/B by SHOWO_GEMM_SK
TAG=${1:-r4h}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x > gpurun_out/${TAG}_kernel_tests.log 2>&1; echo "pytest kernels rc=$?"
grep -E "passed|failed|error" gpurun_out/${TAG}_kernel_tests.log | tail -3; grep -E "^(FAILED|ERROR)|^E  " gpurun_out/${TAG}_kernel_tests.log | head -12
export GEMM_BENCH_VARS=4160,3160,6160 GEMM_BENCH_GNS=4,8
timeout 600 tools/gemm_bench "0:4:0:4160,0:4:0:3160,0:4:0:6160,0:8:0:6160,0:4:0:0" 2 > gpurun_out/${TAG}_harness.txt 2>&1; echo "harness rc=$?"
grep -E "dense|fc2|kcat|4096|8192\^3" gpurun_out/${TAG}_harness.txt | cut -c1-330
for sk in 0 1 0 1; do
  SHOWO_GEMM_SK=$sk timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-train-leg --no-accuracy-leg --no-config-legs 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('t2i SK=$sk', round(d['value'], 2), 'images/s', round(d['ms_per_step'], 1), 'ms  gemm', round(d['roofline']['achieved'], 1), 'TF/s')
"
done
for sk in 0 1 0 1; do
  SHOWO_GEMM_SK=$sk timeout 300 python bench.py --workload train --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('train SK=$sk', round(d['value'], 2), 'ms/step  gemm', round(d['roofline']['achieved'], 1), 'TF/s')
"
done
for sk in 0 1; do
  SHOWO_GEMM_SK=$sk timeout 300 python bench.py --workload t2i512 --steps 2 --warmup 1 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('t2i512 SK=$sk', round(d['value'], 3), 'images/s', round(d['ms_per_step'], 1), 'ms')
"
done
timeout 900 python -m pytest tests/test_modules_gpu.py tests/test_train_gpu.py -m gpu -q -x -k "full_size_t2i or tiny or small_training or two_layer or trainer" 2>&1 | grep -E "passed|failed|error" | tail -2
