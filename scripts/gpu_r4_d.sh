#!/bin/bash
# round-4 pass D: cooperative split-K reduction -- split-K tests, small-M harness shapes and the batch-1 bench, each with SHOWO_GEMM_COOP=0|1
TAG=${1:-r4d}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "splitk or gemm_tn or kcat" > gpurun_out/${TAG}_splitk_tests.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|error" gpurun_out/${TAG}_splitk_tests.log | tail -3; grep -E "^(FAILED|ERROR)|^E  " gpurun_out/${TAG}_splitk_tests.log | head -12
for coop in 0 1; do
  echo "== SHOWO_GEMM_COOP=$coop"
  SHOWO_GEMM_COOP=$coop timeout 300 tools/gemm_bench "0:4:0:0" 4 2>&1 | cut -c1-200 | tee -a gpurun_out/${TAG}_harness_coop$coop.txt
done
for coop in 0 1 0 1; do
  SHOWO_GEMM_COOP=$coop timeout 300 python bench.py --batch 1 --steps 6 --warmup 2 --no-cpu-baseline --no-train-leg --no-accuracy-leg --no-config-legs --roofline-steps 0 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('batch1 COOP=$coop', round(d['value'], 2), 'images/s', round(d['ms_per_step'], 1), 'ms')
"
done
for coop in 0 1; do
  SHOWO_GEMM_COOP=$coop timeout 300 python bench.py --workload mmu --steps 1 --warmup 1 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('mmu COOP=$coop', round(d['value'], 1), 'tokens/s  prefill->first token', round(d['config']['prefill_to_first_token_ms'], 2), 'ms  clip+proj', round(d['config']['clip_projector_splice_ms'], 2), 'ms')
"
done
timeout 900 python -m pytest tests/test_modules_gpu.py tests/test_clip_gpu.py -m gpu -q -x -k "cfg4 or tiny_mmu or clip" 2>&1 | grep -E "passed|failed|error" | tail -2
