#!/bin/bash
# round-2 GPU pass P: co-scheduled decode layer (attention || fc2 in one launch): decode tests under both modes, cfg4 A/B
TAG=${1:-r2p}
mkdir -p gpurun_out
for m in 0 2; do
echo "decode tests, SHOWO_DECODE_FORK=$m"
SHOWO_DECODE_FORK=$m timeout 600 python -m pytest tests -m gpu -x -q -k "decode or mmu or generate" 2>&1 | tail -3
done
for v in "0 224" "2 224" "2 128" "2 96" "0 224" "2 224"; do
set -- $v
SHOWO_DECODE_FORK=$1 SHOWO_DECODE_CO_BLOCKS=$2 timeout 300 python bench.py --workload mmu --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/bench_mmu_${TAG}_$1_$2.log 2>&1
echo "fork=$1 co_blocks=$2"; grep -h '"metric"' gpurun_out/bench_mmu_${TAG}_$1_$2.log | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(round(d['value'],1), 'tok/s', round(d['roofline']['achieved']), 'GB/s', d['config'].get('ms_per_decoded_token'))
"
done
