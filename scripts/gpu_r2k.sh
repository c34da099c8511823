#!/bin/bash
# round-2 GPU pass K: forked decode layer A/B (cfg4), decode parity tests
TAG=${1:-r2k}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_modules_gpu.py tests/test_kernels_gpu.py -m gpu -x -q -k "mmu or decode or argmax or sample_topk or kv_cache" 2>&1 | tail -4
for v in 1 0 1 0; do
SHOWO_DECODE_FORK=$v timeout 300 python bench.py --workload mmu --steps 1 --warmup 1 > gpurun_out/bench_mmu_${TAG}_fork$v.log 2>&1
echo "fork=$v rc=$?"; grep '"metric"' gpurun_out/bench_mmu_${TAG}_fork$v.log | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(round(d['value'],1), 'tok/s', round(d['config']['ms_per_decoded_token'],4), 'ms/token', round(d['roofline']['achieved']), 'GB/s', 'ttft', round(d['config']['time_to_first_token_ms'],2))
"; grep -E "Error|error" gpurun_out/bench_mmu_${TAG}_fork$v.log | head -3
done
