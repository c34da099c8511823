#!/bin/bash
# round 5, pass O: kernel table of the batched decode WITHOUT the co-scheduled launch (attention alone, dense + fc2 in one launch)
R=$(pwd)
mkdir -p gpurun_out/r5o
export TMPDIR=/tmp
cd /tmp
SHOWO_DECODE_BATCH_CO=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r5o/prof -o prof -- python $R/bench.py --workload mmu --steps 1 --warmup 1 > $R/gpurun_out/r5o/prof.log 2>&1
cd $R
find gpurun_out/r5o/prof -type f ! -name "*stats*" -size +2M -delete
f=$(find gpurun_out/r5o/prof -name "*kernel_stats.csv" | head -1); grep -E "gemvB|attn_decode_kernel|y2B|seam_rows" $f | cut -c1-150
