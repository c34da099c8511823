#!/bin/bash
# round-4 pass J: kernel table of the cfg3 (512x512 inpainting) bench
R=$(pwd); mkdir -p gpurun_out; export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r4j_t2i512 -o prof -- python $R/bench.py --workload t2i512 --steps 1 --warmup 1 > $R/gpurun_out/prof_r4j_t2i512.log 2>&1
cd $R; find gpurun_out/prof_r4j_t2i512 -type f ! -name "*stats*" -size +2M -delete
head -22 gpurun_out/prof_r4j_t2i512/prof_kernel_stats.csv | cut -c1-190
