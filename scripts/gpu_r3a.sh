#!/bin/bash
# round-2 GPU pass 3A: tile-group width sweep of the fused layer GEMMs (harness, rotating weights) + the tuner's full candidate table in the bench
TAG=${1:-r3a}
mkdir -p gpurun_out
GEMM_BENCH_VARS=3192,3160,2240 GEMM_BENCH_GNS=2,4,7,8,14,28,56 timeout 300 tools/gemm_bench "5:8:20:3192" 2 2>&1 | grep -E "^kcat|^qkv" | tee gpurun_out/gemm_${TAG}_gn.log
SHOWO_GEMM_TUNE_LOG=2 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_${TAG}.log 2>&1
grep "tune\]" gpurun_out/bench_${TAG}.log | grep -A1 "M=4128" | cut -c1-400
grep -h '"metric"' gpurun_out/bench_${TAG}.log | cut -c1-120
