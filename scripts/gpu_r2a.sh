#!/bin/bash
# round-2 GPU pass A: GEMM variant harness, parity tests, ceilings, fused-vs-unfused bench A/B.  Outputs under gpurun_out/.
TAG=${1:-r2a}
mkdir -p gpurun_out
V="5:8:0:256,5:8:0:240,5:8:0:224,5:8:0:208,5:8:0:176,5:8:0:160,5:8:0:144,5:8:0:1192,5:8:0:1176,5:8:0:1160,5:8:0:1144,5:8:0:1128"
timeout 600 tools/gemm_bench $V 2 > gpurun_out/gemm_$TAG.log 2>&1; echo "gemm_bench rc=$?" >> gpurun_out/gemm_$TAG.log
tail -8 gpurun_out/gemm_$TAG.log | cut -c1-300
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
