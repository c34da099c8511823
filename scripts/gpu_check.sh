#!/bin/bash
# One GPU-box pass: parity tests, smoke, bench line, rocprofv3 kernel stats.  Outputs under gpurun_out/.
# usage: scripts/gpu_check.sh [tag]
TAG=${1:-r1}
R=$(pwd)
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -rA 2>&1 | tail -80 > gpurun_out/tests_$TAG.log
echo "pytest rc=${PIPESTATUS[0]}" >> gpurun_out/tests_$TAG.log
tail -5 gpurun_out/tests_$TAG.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke_$TAG.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke_$TAG.log
tail -2 gpurun_out/smoke_$TAG.log
timeout 600 python bench.py --steps 3 --warmup 1 > gpurun_out/bench_$TAG.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench_$TAG.log
tail -4 gpurun_out/bench_$TAG.log
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o prof -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_$TAG.log 2>&1
echo "rocprof rc=$?" >> $R/gpurun_out/prof_$TAG.log
cd $R
find gpurun_out/prof_$TAG -name "*kernel_stats*" | head
# keep only the small summaries (the full trace is large)
find gpurun_out/prof_$TAG -type f ! -name "*stats*" -size +2M -delete
ls -la gpurun_out/prof_$TAG/* | head -20
