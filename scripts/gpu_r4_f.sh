#!/bin/bash
# round-4 pass F: the phase-split conv (with split-K) for small launches too -- VQ tests, then batch-1 / mmu / t2i512 A/B by SHOWO_CONV_SPLIT_MINM
TAG=${1:-r4f}
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -k "magvit or groupnorm or conv or quantizer" 2>&1 | grep -E "passed|failed|error" | tail -3
for mm in 2048 256 2048 256; do
  SHOWO_CONV_SPLIT_MINM=$mm timeout 300 python bench.py --batch 1 --steps 6 --warmup 2 --no-cpu-baseline --no-train-leg --no-accuracy-leg --no-config-legs --roofline-steps 1 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('batch1 MINM=$mm', round(d['value'], 2), 'images/s', round(d['ms_per_step'], 1), 'ms  vq_conv', round(d['roofline']['vq_conv']['achieved'], 1))
"
done
for mm in 2048 256; do
  SHOWO_CONV_SPLIT_MINM=$mm timeout 300 python bench.py --workload t2i512 --steps 2 --warmup 1 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('t2i512 MINM=$mm', round(d['value'], 3), 'images/s', round(d['ms_per_step'], 1), 'ms')
"
done
