#!/bin/bash
# round-2 GPU pass N: LDS-staged bf16 epilogue stores: harness decomposition A/B, GEMM tests, pipeline A/B
TAG=${1:-r2n}
mkdir -p gpurun_out
timeout 300 tools/gemm_bench "5:8:12:3192,5:8:20:3192,5:8:12:256,5:8:20:256" 3 2>&1 | sed -E "s/tiles256= *[0-9]+//" | cut -c1-200 | tee gpurun_out/gemm_${TAG}_decomp.log
timeout 400 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "gemm or qkv or kcat" 2>&1 | tail -3
export SHOWO_GEMM_TUNE_LOG=1
for v in 1 0 1 0; do
SHOWO_GEMM_STAGE=$v timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/bench_${TAG}_stage$v.log 2>&1
echo "stage=$v"; grep "tune\]" gpurun_out/bench_${TAG}_stage$v.log | grep "M=4128" | cut -c1-100; grep -h '"metric"' gpurun_out/bench_${TAG}_stage$v.log | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(round(d['value'],2), 'img/s', round(d['roofline']['achieved']), 'TF/s gemm', d['roofline']['avg_launch_ms'])
"
done
