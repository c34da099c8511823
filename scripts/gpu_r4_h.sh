#!/bin/bash
# round-4 pass H (second attempt; the first one hung on a HOST-side self-deadlock of the dispatch mutex, not on the GPU): stream-K form of the
# production GEMM -- its tests first (the script stops there on any failure), then same-box A/B of the t2i and training benches.  Tight timeouts.
TAG=${1:-r4h}
mkdir -p gpurun_out
timeout 240 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "stream_k or kcat" > gpurun_out/${TAG}_sk_tests.log 2>&1; rc=$?; echo "pytest stream-K rc=$rc"
grep -E "passed|failed|error" gpurun_out/${TAG}_sk_tests.log | tail -3; grep -E "^(FAILED|ERROR)|^E  " gpurun_out/${TAG}_sk_tests.log | head -12
if [ $rc -ne 0 ]; then echo "stopping: stream-K tests failed"; exit 1; fi
for sk in 1 0 1 0; do
  SHOWO_GEMM_SK=$sk timeout 120 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-train-leg --no-accuracy-leg --no-config-legs 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('t2i SK=$sk', round(d['value'], 2), 'images/s', round(d['ms_per_step'], 1), 'ms  gemm', round(d['roofline']['achieved'], 1), 'TF/s')
"
done
for sk in 1 0; do
  SHOWO_GEMM_SK=$sk timeout 150 python bench.py --workload train --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('train SK=$sk', round(d['value'], 2), 'ms/step  gemm', round(d['roofline']['achieved'], 1), 'TF/s')
"
done
