#!/bin/bash
# rocprofv3 kernel stats of the AR decode (tools/mmu_prof.py: prefill of 631 ids + 33 decode steps, twice) + the cfg3/cfg4 lines
TAG=${1:-r1}
R=$(pwd)
mkdir -p gpurun_out
python tools/run_configs.py 2>&1 | grep "^cfg" > gpurun_out/configs_$TAG.log
cat gpurun_out/configs_$TAG.log
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_decode_$TAG -o prof -- python $R/tools/mmu_prof.py > $R/gpurun_out/prof_decode_$TAG.log 2>&1
cd $R
find gpurun_out/prof_decode_$TAG -type f ! -name "*stats*" -size +1M -delete
find gpurun_out/prof_decode_$TAG -name "*kernel_stats*" | head -2
