#!/bin/bash
# round-2 GPU pass Q: decode attention loads no longer wait for pos_dev; co-scheduled default; full-size decode form test; cfg4 A/B
TAG=${1:-r2q}
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "decode or mmu or generate" 2>&1 | tail -3
for v in "2 96" "2 64" "0 96" "2 96" "2 48"; do
set -- $v
SHOWO_DECODE_FORK=$1 SHOWO_DECODE_CO_BLOCKS=$2 timeout 300 python bench.py --workload mmu --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/bench_mmu_${TAG}_$1_$2.log 2>&1
echo "fork=$1 co_blocks=$2"; grep -h '"metric"' gpurun_out/bench_mmu_${TAG}_$1_$2.log | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(round(d['value'],1), 'tok/s', round(d['roofline']['achieved']), 'GB/s', d['config'].get('ms_per_decoded_token'))
"
done
