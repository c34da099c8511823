#!/bin/bash
# round-2 GPU pass X: split-K of launches with few tiles: kernel tests, cfg4 A/B (prefill + CLIP), module tests
TAG=${1:-r2x}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "splitk or kcat or gemm_epilogues" 2>&1 | tail -4
for v in 1 0 1 0; do
SHOWO_GEMM_SPLITK=$v timeout 300 python bench.py --workload mmu --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/bench_mmu_${TAG}_sk$v.log 2>&1
echo "splitk=$v"; grep -h '"metric"' gpurun_out/bench_mmu_${TAG}_sk$v.log | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); c = d['config']; print(round(d['value'],1), 'tok/s', 'clip', round(c['clip_projector_splice_ms'],2), 'prefill', round(c['prefill_to_first_token_ms'],2), 'ttft', round(c['time_to_first_token_ms'],2), 'ms/token', round(c['ms_per_decoded_token'],4))
"
done
timeout 1200 python -m pytest tests/test_modules_gpu.py tests/test_clip_gpu.py -m gpu -x -q 2>&1 | tail -4
