#!/bin/bash
# round-2 GPU pass R: Infinity-Cache warm-up of the next GEMM's weights on a side stream (SHOWO_MALL_PF bits), t2i A/B
TAG=${1:-r2r}
mkdir -p gpurun_out
for v in "0 64" "3 64" "1 64" "2 64" "3 32" "0 64" "3 128"; do
set -- $v
SHOWO_MALL_PF=$1 SHOWO_MALL_PF_BLOCKS=$2 timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/bench_${TAG}_pf$1_$2.log 2>&1
echo "mall_pf=$1 blocks=$2"; grep -h '"metric"' gpurun_out/bench_${TAG}_pf$1_$2.log | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(round(d['value'],2), 'img/s', round(d['roofline']['achieved']), 'TF/s gemm', d['roofline']['avg_launch_ms'])
"
done
