#!/bin/bash
# HBM-side traffic of the bench command with the final build: FETCH_SIZE and WRITE_SIZE, one rocprofv3 --pmc pass each
TAG=${1:-r2u}
R=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
i=0
for G in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  timeout 400 rocprofv3 --pmc $G --output-format csv -d /tmp/pmc_$i -o pmc -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-train-leg --no-accuracy-leg --no-config-legs --roofline-steps 0 --no-events > $R/gpurun_out/pmc_${TAG}_$i.log 2>&1
  echo "pmc pass $i ($G) rc=$?"
done
cd $R
python tools/pmc_summary.py $TAG /tmp/pmc_1 /tmp/pmc_2 gpurun_out/pmc_${TAG}_traffic.json
