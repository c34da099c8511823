#!/bin/bash
# round-4 pass G: write-through (sc1) partial stores in the cooperative split-K reduction: tests + small-M harness + batch-1 bench by SHOWO_GEMM_COOP=1|2
mkdir -p gpurun_out
for coop in 1 2; do
  SHOWO_GEMM_COOP=$coop timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "splitk or kcat" 2>&1 | grep -E "passed|failed|error" | tail -2
done
for coop in 1 2; do
  echo "== SHOWO_GEMM_COOP=$coop"; SHOWO_GEMM_COOP=$coop timeout 300 tools/gemm_bench "0:4:0:0" 4 2>&1 | cut -c1-160 | grep -E "fc2|dense|out"
done
for coop in 1 2 1 2; do
  SHOWO_GEMM_COOP=$coop timeout 300 python bench.py --batch 1 --steps 6 --warmup 2 --no-cpu-baseline --no-train-leg --no-accuracy-leg --no-config-legs --roofline-steps 0 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('batch1 COOP=$coop', round(d['value'], 2), 'images/s', round(d['ms_per_step'], 1), 'ms')
"
done
