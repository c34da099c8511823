#!/bin/bash
# round-2 GPU pass 3B: per-row-block k rotation in the ring GEMMs (SHOWO_GEMM_KROT): harness with rotating weights + pipeline A/B
TAG=${1:-r3b}
mkdir -p gpurun_out
for k in 0 1; do
echo "KROT=$k"
SHOWO_GEMM_KROT=$k GEMM_BENCH_VARS=3192,3160,4160,2240 GEMM_BENCH_GNS=4,8 timeout 300 tools/gemm_bench "5:8:20:3192" 2 2>&1 | grep -E "^kcat M=(4128|6192)|^qkv\|fc1 M=(4128|6192)" | tee -a gpurun_out/gemm_${TAG}_krot.log
done
for k in 0 1 0 1; do
SHOWO_GEMM_KROT=$k timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/bench_${TAG}_krot$k.log 2>&1
echo "krot=$k"; grep -h '"metric"' gpurun_out/bench_${TAG}_krot$k.log | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(round(d['value'],2), 'img/s', round(d['roofline']['achieved']), 'TF/s gemm', d['roofline']['avg_launch_ms'])
"
done
SHOWO_GEMM_KROT=1 timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "gemm or qkv or kcat" 2>&1 | tail -2
