#!/bin/bash
# HBM traffic of the bench kernels from the PMC counters: one rocprofv3 --pmc pass per counter (TCC fits one at a time),
# counters only (no sys/hip/hsa tracing in the same run).  usage: scripts/gpu_pmc.sh [tag]
TAG=${1:-r1}
R=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_$C -o pmc -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-events > $R/gpurun_out/pmc_${TAG}_$C.log 2>&1
  echo "pmc $C rc=$?"
done
cd $R
python tools/pmc_summary.py $TAG /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE gpurun_out/pmc_${TAG}_traffic.json
