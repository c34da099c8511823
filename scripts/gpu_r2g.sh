#!/bin/bash
# round-2 GPU pass G: new parity tests, train bench (roofline + cpu baseline), cfg3 / cfg4 workloads, decode profile.
TAG=${1:-r2g}
R=$(pwd)
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -s 2>&1 | grep -E "parity|passed|failed|Error|assert|FAILED|skipped" | cut -c1-260 > gpurun_out/tests_$TAG.log
grep -E "rounding-point|per-block|passed|failed|FAILED|Error" gpurun_out/tests_$TAG.log | tail -22
timeout 900 python bench.py --workload train --steps 3 --warmup 1 > gpurun_out/bench_train_$TAG.log 2>&1; echo "train rc=$?" >> gpurun_out/bench_train_$TAG.log
grep -E '"metric"|rc=|Error' gpurun_out/bench_train_$TAG.log | cut -c1-2500
timeout 600 python bench.py --workload t2i512 --steps 2 --warmup 1 > gpurun_out/bench_t2i512_$TAG.log 2>&1; echo "t2i512 rc=$?" >> gpurun_out/bench_t2i512_$TAG.log
grep -E '"metric"|rc=|Error' gpurun_out/bench_t2i512_$TAG.log | cut -c1-1600
timeout 600 python bench.py --workload mmu --steps 1 --warmup 1 > gpurun_out/bench_mmu_$TAG.log 2>&1; echo "mmu rc=$?" >> gpurun_out/bench_mmu_$TAG.log
grep -E '"metric"|rc=|Error' gpurun_out/bench_mmu_$TAG.log | cut -c1-1600
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_mmu_$TAG -o prof -- python $R/bench.py --workload mmu --steps 1 --warmup 1 > $R/gpurun_out/prof_mmu_$TAG.log 2>&1
cd $R
find gpurun_out/prof_mmu_$TAG -type f ! -name "*stats*" -size +2M -delete
head -12 gpurun_out/prof_mmu_$TAG/prof_kernel_stats.csv | cut -c1-160
