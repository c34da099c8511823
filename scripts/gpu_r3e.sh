#!/bin/bash
# round-2 GPU pass 3E: split-K floor (k-tiles per split) 16 vs 8 on cfg4
TAG=${1:-r3e}
mkdir -p gpurun_out
for v in 16 8 16 8 4; do
SHOWO_GEMM_SPLITK_MIN=$v timeout 300 python bench.py --workload mmu --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/bench_mmu_${TAG}_min$v.log 2>&1
echo "splitk_min=$v"; grep -h '"metric"' gpurun_out/bench_mmu_${TAG}_min$v.log | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); c = d['config']; print(round(d['value'],1), 'tok/s', 'clip', round(c['clip_projector_splice_ms'],2), 'prefill', round(c['prefill_to_first_token_ms'],2), 'ttft', round(c['time_to_first_token_ms'],2), 'ms/token', round(c['ms_per_decoded_token'],4))
"
done
