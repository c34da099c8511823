#!/bin/bash
# round-2 GPU pass B: parity tests, fused-vs-unfused bench A/B, ceilings.  Outputs under gpurun_out/.
TAG=${1:-r2b}
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > gpurun_out/tests_$TAG.log
tail -3 gpurun_out/tests_$TAG.log
export SHOWO_GEMM_TUNE_LOG=1
timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/bench_${TAG}_fused.log 2>&1; echo "rc=$?" >> gpurun_out/bench_${TAG}_fused.log
SHOWO_FUSED_LAYER=0 timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/bench_${TAG}_unfused.log 2>&1; echo "rc=$?" >> gpurun_out/bench_${TAG}_unfused.log
timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/bench_${TAG}_fused2.log 2>&1; echo "rc=$?" >> gpurun_out/bench_${TAG}_fused2.log
grep -h '"metric"' gpurun_out/bench_${TAG}_*.log | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(round(d['value'],2), 'img/s', round(d['roofline']['achieved']), 'TF/s', d['roofline']['avg_launch_ms'])
"
unset SHOWO_GEMM_TUNE_LOG
timeout 600 python tools/ceiling.py gpurun_out/ceiling_$TAG.json > gpurun_out/ceiling_$TAG.log 2>&1; echo "rc=$?" >> gpurun_out/ceiling_$TAG.log
tail -5 gpurun_out/ceiling_$TAG.log | cut -c1-300
