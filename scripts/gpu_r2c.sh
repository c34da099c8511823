#!/bin/bash
# round-2 GPU pass C: weight-prefetch A/B in the harness (cold weights) and in the pipeline; rocprof kernel stats.
TAG=${1:-r2c}
R=$(pwd)
mkdir -p gpurun_out
V="5:8:4:256,5:8:2:256,5:8:4:240,5:8:2:240,5:8:4:208,5:8:2:208,5:8:4:1176,5:8:2:1176,5:8:4:1160,5:8:2:1160,5:8:4:1144,5:8:2:1144"
timeout 600 tools/gemm_bench $V 2 > gpurun_out/gemm_$TAG.log 2>&1; echo "gemm_bench rc=$?" >> gpurun_out/gemm_$TAG.log
grep -E "kcat|qkv\|fc1 M|MISMATCH" gpurun_out/gemm_$TAG.log | cut -c1-250
export SHOWO_GEMM_TUNE_LOG=1
timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/bench_${TAG}_pf1.log 2>&1; echo "rc=$?" >> gpurun_out/bench_${TAG}_pf1.log
SHOWO_GEMM_PF=0 timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/bench_${TAG}_pf0.log 2>&1; echo "rc=$?" >> gpurun_out/bench_${TAG}_pf0.log
timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/bench_${TAG}_pf1b.log 2>&1; echo "rc=$?" >> gpurun_out/bench_${TAG}_pf1b.log
grep -h '"metric"' gpurun_out/bench_${TAG}_*.log | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(round(d['value'],2), 'img/s', round(d['roofline']['achieved']), 'TF/s', d['roofline']['avg_launch_ms'])
"
unset SHOWO_GEMM_TUNE_LOG
timeout 300 python -m pytest tests/test_kernels_gpu.py tests/test_modules_gpu.py -m gpu -x -q 2>&1 | tail -3
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o prof -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_$TAG.log 2>&1
echo "rocprof rc=$?" >> $R/gpurun_out/prof_$TAG.log
cd $R
find gpurun_out/prof_$TAG -type f ! -name "*stats*" -size +2M -delete
find gpurun_out/prof_$TAG -name "*kernel_stats*" | head -3
