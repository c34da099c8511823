#!/bin/bash
# round-2 GPU pass J: pipeline A/B ring on/off in one box + rocprof kernel stats with the ring kernels
TAG=${1:-r2j}
R=$(pwd)
mkdir -p gpurun_out
export SHOWO_GEMM_TUNE_LOG=1
for i in 1 2; do
timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/bench_${TAG}_ring1_$i.log 2>&1
SHOWO_GEMM_RING=0 timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/bench_${TAG}_ring0_$i.log 2>&1
done
for f in gpurun_out/bench_${TAG}_ring*.log; do echo $f; grep "tune\]" $f | grep "M=4128" | cut -c1-100; grep -h '"metric"' $f | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(round(d['value'],2), 'img/s', round(d['roofline']['achieved']), 'TF/s', d['roofline']['avg_launch_ms'])
"; done
unset SHOWO_GEMM_TUNE_LOG
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o prof -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --roofline-steps 0 > $R/gpurun_out/prof_$TAG.log 2>&1
cd $R
find gpurun_out/prof_$TAG -type f ! -name "*stats*" -size +2M -delete
head -8 gpurun_out/prof_$TAG/prof_kernel_stats.csv | cut -c1-170
