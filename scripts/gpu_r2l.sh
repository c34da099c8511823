#!/bin/bash
# round-2 GPU pass L: attention with five query tiles per block: parity tests + pipeline A/B
TAG=${1:-r2l}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_modules_gpu.py tests/test_train_gpu.py tests/test_clip_gpu.py -m gpu -x -q 2>&1 | tail -4
for v in 1 0 1 0; do
SHOWO_ATTN_WPB5=$v timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/bench_${TAG}_wpb$v.log 2>&1
echo "wpb5=$v"; grep -h '"metric"' gpurun_out/bench_${TAG}_wpb$v.log | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(round(d['value'],2), 'img/s', round(d['roofline']['achieved']), 'TF/s gemm', round(d['roofline']['attention']['achieved']), 'TF/s attn')
"
done
