#!/bin/bash
# round-4 pass B: the four-wave GEMM (variant 5256) in the harness (race screen against the 128^2 kernel, TF/s on the pipeline
# shapes with cold weights, fused entry points), the GEMM bit-identity tests with the variant forced, the per-block gate with the
# tiled-attention rounding model, and a same-box A/B of the headline bench
TAG=${1:-r4b}
mkdir -p gpurun_out
export GEMM_BENCH_VARS=3192,2256,5256 GEMM_BENCH_GNS=4,8
timeout 600 tools/gemm_bench "0:4:0:3192,0:4:0:2256,0:4:0:5256,0:8:0:5256" 2 > gpurun_out/${TAG}_harness.txt 2>&1; echo "harness rc=$?"
cat gpurun_out/${TAG}_harness.txt | cut -c1-330
SHOWO_GEMM_BM=5256 timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "gemm or qkv or kcat" > gpurun_out/${TAG}_gemm_tests_5256.log 2>&1; echo "pytest(5256 forced) rc=$?"
grep -E "passed|failed|error" gpurun_out/${TAG}_gemm_tests_5256.log | tail -3; grep -E "^(FAILED|ERROR)|^E  " gpurun_out/${TAG}_gemm_tests_5256.log | head -12
SHOWO_GEMM_BM=5256 timeout 900 python -m pytest tests/test_modules_gpu.py -m gpu -q -x -rP -k "tiny_forward or full_size_logits or rows_equals" > gpurun_out/${TAG}_mod_tests_5256.log 2>&1; echo "pytest modules(5256 forced) rc=$?"
grep -E "passed|failed|error" gpurun_out/${TAG}_mod_tests_5256.log | tail -3; grep -E "^\[parity\].*(per-block|full-size logits)" gpurun_out/${TAG}_mod_tests_5256.log | cut -c1-420; grep -E "^(FAILED|ERROR)|^E  " gpurun_out/${TAG}_mod_tests_5256.log | head -12
for bm in 0 5256 0 5256; do
  if [ $bm = 0 ]; then unset SHOWO_GEMM_BM; else export SHOWO_GEMM_BM=$bm; fi
  timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-train-leg --no-accuracy-leg --no-config-legs 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('bench BM=$bm', round(d['value'], 2), 'images/s', round(d['ms_per_step'], 1), 'ms  gemm', round(d['roofline']['achieved'], 1), 'TF/s')
"
done
unset SHOWO_GEMM_BM
SHOWO_GEMM_TUNE_LOG=2 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train-leg --no-accuracy-leg --no-config-legs 2>&1 | grep "gemm2p tune" | cut -c1-400
