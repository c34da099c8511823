#!/bin/bash
# round-2 GPU pass H: 3-deep weight ring variants (gemm3w) in the harness (cold weights), GEMM tests, pipeline A/B.
TAG=${1:-r2h}
mkdir -p gpurun_out
timeout 600 tools/gemm_bench "5:8:4:3192,5:8:4:4192,5:8:4:3160,5:8:4:4160,5:8:4:3144,5:8:4:4144" 2 > gpurun_out/gemm_$TAG.log 2>&1; echo "gemm_bench rc=$?" >> gpurun_out/gemm_$TAG.log
grep -E "kcat|qkv\|fc1 M|MISMATCH|rc=" gpurun_out/gemm_$TAG.log | cut -c1-250
grep -E "act258|full|4096|8192" gpurun_out/gemm_$TAG.log | sed -E 's/ TF +[0-9.]+ ms//g' | cut -c1-230
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "gemm or qkv or kcat" 2>&1 | tail -3
export SHOWO_GEMM_TUNE_LOG=1
timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/bench_${TAG}.log 2>&1; echo "rc=$?" >> gpurun_out/bench_${TAG}.log
grep "tune\]" gpurun_out/bench_${TAG}.log | head -8
grep -h '"metric"' gpurun_out/bench_${TAG}.log | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(round(d['value'],2), 'img/s', round(d['roofline']['achieved']), 'TF/s', d['roofline']['avg_launch_ms'])
"
