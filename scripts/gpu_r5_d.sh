#!/bin/bash
# round 5, pass D: kernel table of the mmu bench (batch 1 + batch 4), dense-mask diagnostic of the split attention
R=$(pwd)
mkdir -p gpurun_out/r5d
timeout 300 python tools/diag_split_attn.py > gpurun_out/r5d/diag_attn.log 2>&1; cat gpurun_out/r5d/diag_attn.log | tail -8
export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r5d/prof_mmu -o prof -- python $R/bench.py --workload mmu --steps 1 --warmup 1 > $R/gpurun_out/r5d/prof_mmu.log 2>&1
cd $R
find gpurun_out/r5d/prof_mmu -type f ! -name "*stats*" -size +2M -delete
f=$(find gpurun_out/r5d/prof_mmu -name "*kernel_stats.csv" | head -1); head -30 $f | cut -c1-160
