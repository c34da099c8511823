#!/bin/bash
# round-2 GPU pass D: tiled-weight A/B in the harness (cold weights) and in the pipeline; counter list.
TAG=${1:-r2d}
R=$(pwd)
mkdir -p gpurun_out
timeout 600 tools/gemm_bench "5:8:4:256" 2 > gpurun_out/gemm_$TAG.log 2>&1; echo "gemm_bench rc=$?" >> gpurun_out/gemm_$TAG.log
grep -E "kcat|qkv\|fc1 M|MISMATCH" gpurun_out/gemm_$TAG.log | cut -c1-250
timeout 300 python -m pytest tests/test_kernels_gpu.py tests/test_modules_gpu.py -m gpu -x -q 2>&1 | tail -3
export SHOWO_GEMM_TUNE_LOG=1
export SHOWO_GEMM_PF=0
timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/bench_${TAG}_t1.log 2>&1; echo "rc=$?" >> gpurun_out/bench_${TAG}_t1.log
SHOWO_W_TILED=0 timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/bench_${TAG}_t0.log 2>&1; echo "rc=$?" >> gpurun_out/bench_${TAG}_t0.log
timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/bench_${TAG}_t1b.log 2>&1; echo "rc=$?" >> gpurun_out/bench_${TAG}_t1b.log
grep -h '"metric"' gpurun_out/bench_${TAG}_*.log | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(round(d['value'],2), 'img/s', round(d['roofline']['achieved']), 'TF/s', d['roofline']['avg_launch_ms'])
"
rocprofv3 -L > gpurun_out/counters_list.txt 2>&1
wc -l gpurun_out/counters_list.txt
