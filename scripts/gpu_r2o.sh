#!/bin/bash
# round-2 GPU pass O: V^T stores transposed through LDS (lanes along pos): QKV tests, pipeline A/B (SHOWO_GEMM_STAGE=3 keeps V^T direct)
TAG=${1:-r2o}
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "gemm or qkv or kcat" 2>&1 | tail -3
export SHOWO_GEMM_TUNE_LOG=1
for v in 1 3 1 3; do
SHOWO_GEMM_STAGE=$v timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/bench_${TAG}_stage$v.log 2>&1
echo "stage=$v"; grep "tune\]" gpurun_out/bench_${TAG}_stage$v.log | grep "M=4128" | cut -c1-100; grep -h '"metric"' gpurun_out/bench_${TAG}_stage$v.log | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(round(d['value'],2), 'img/s', round(d['roofline']['achieved']), 'TF/s gemm', d['roofline']['avg_launch_ms'])
"
done
