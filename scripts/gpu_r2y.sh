#!/bin/bash
# round-2 GPU pass Y: decode layer seam (dense GEMV + next layer's LN/qkv/fc1 GEMV in one launch with a grid barrier): tests, cfg4 A/B
TAG=${1:-r2y}
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q -k "decode or mmu or generate or splitk" 2>&1 | tail -3
for v in 1 0 1 0; do
SHOWO_DECODE_SEAM=$v timeout 300 python bench.py --workload mmu --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/bench_mmu_${TAG}_seam$v.log 2>&1
echo "seam=$v"; grep -h '"metric"' gpurun_out/bench_mmu_${TAG}_seam$v.log | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); c = d['config']; print(round(d['value'],1), 'tok/s', 'ttft', round(c['time_to_first_token_ms'],2), 'ms/token', round(c['ms_per_decoded_token'],4))
"
tail -2 gpurun_out/bench_mmu_${TAG}_seam$v.log | grep -i -E "error|timed" | head -2
done
