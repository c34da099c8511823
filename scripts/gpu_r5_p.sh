#!/bin/bash
# round 5, pass P: Infinity-Cache prefetch role of the decode layers -- sweep in one process, then the kernel table at the full setting
R=$(pwd)
mkdir -p gpurun_out/r5p
export TMPDIR=/tmp
timeout 600 python tools/decode_sweep.py --configs "pf=0:0:0;pf=0:1:64;pf=16:1:64;pf=32:1:64;pf=58:1:64;pf=58:0:64;pf=58:1:32;pf=58:1:96;pf=32:1:96" > gpurun_out/r5p/sweep.txt 2> gpurun_out/r5p/sweep.err
cat gpurun_out/r5p/sweep.txt; tail -3 gpurun_out/r5p/sweep.err
cd /tmp
for st in 0:0:0 58:1:64; do
  tag=$(echo $st | tr ':' '_')
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r5p/prof_$tag -o prof -- python $R/tools/decode_sweep.py --configs pf=$st --reps 1 > $R/gpurun_out/r5p/prof_$tag.log 2>&1
  f=$(find $R/gpurun_out/r5p/prof_$tag -name "*kernel_stats.csv" | head -1)
  echo "== $st"; grep -E "gemv|attn_decode|y2|seam" $f | cut -c1-160
done
cd $R
find gpurun_out/r5p -type f ! -name "*stats*" -size +2M -delete
