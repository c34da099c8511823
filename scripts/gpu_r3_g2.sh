#!/bin/bash
# round-3 GPU pass G2: untracked prefetch DMAs in the three attention kernels -- suite, t2i bench (attention TF/s), training step
TAG=${1:-r3g2}
mkdir -p gpurun_out
bash scripts/gpu_tests.sh $TAG
for i in 1 2; do
timeout 300 python bench.py --steps 6 --warmup 2 --no-train-leg --no-cpu-baseline > gpurun_out/${TAG}_t2i.log 2>&1
grep -h '"metric"' gpurun_out/${TAG}_t2i.log | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d['roofline']; print(round(d['value'], 2), 'img/s  gemm', round(r['achieved']), 'TF/s  attention', r['attention'])
"
done
timeout 400 python bench.py --workload train --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_train.log 2>&1
grep -h '"metric"' gpurun_out/${TAG}_train.log | tail -1 | cut -c 1-160
