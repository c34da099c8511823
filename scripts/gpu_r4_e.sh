#!/bin/bash
# round-4 pass E: split-K of the split-precision conv (VQGAN 16x16 / 32x32 levels) -- VQ parity tests, then A/B of the t2i and training benches;
# the new 24-layer gradient test
TAG=${1:-r4e}
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -rP -k "magvit or groupnorm or conv or quantizer or lfq" > gpurun_out/${TAG}_vq_tests.log 2>&1; echo "pytest vq rc=$?"
grep -E "passed|failed|error" gpurun_out/${TAG}_vq_tests.log | tail -3; grep -E "^\[parity\] magvit" gpurun_out/${TAG}_vq_tests.log | cut -c1-200; grep -E "^(FAILED|ERROR)|^E  " gpurun_out/${TAG}_vq_tests.log | head -12
timeout 900 python -m pytest tests/test_train_gpu.py -m gpu -q -x -rP -k "24_layer" > gpurun_out/${TAG}_grad24_test.log 2>&1; echo "pytest 24-layer rc=$?"
grep -E "passed|failed|error" gpurun_out/${TAG}_grad24_test.log | tail -3; grep -E "^\[parity\] full-size 24" gpurun_out/${TAG}_grad24_test.log | cut -c1-1500; grep -E "^(FAILED|ERROR)|^E  " gpurun_out/${TAG}_grad24_test.log | head -12
for sk in 0 1 0 1; do
  SHOWO_CONV_SPLITK=$sk timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-train-leg --no-accuracy-leg --no-config-legs 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('t2i CONV_SPLITK=$sk', round(d['value'], 2), 'images/s', round(d['ms_per_step'], 1), 'ms  vq_conv', round(d['roofline']['vq_conv']['achieved'], 1), 'TF/s')
"
done
for sk in 0 1 0 1; do
  SHOWO_CONV_SPLITK=$sk timeout 300 python bench.py --workload train --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('train CONV_SPLITK=$sk', round(d['value'], 2), 'ms/step  vq_conv', round(d['roofline']['vq_conv']['achieved'], 1), 'TF/s')
"
done
