#!/bin/bash
# PMC passes over the bench command (counters only, one rocprofv3 run per group; TCC has 4 slots, FETCH_SIZE takes 3, WRITE_SIZE 2).
# usage: scripts/gpu_pmc2.sh [tag]
TAG=${1:-r2}
R=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
i=0
for G in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_LEVEL_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  timeout 400 rocprofv3 --pmc $G --output-format csv -d /tmp/pmc_$i -o pmc -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-events > $R/gpurun_out/pmc_${TAG}_$i.log 2>&1
  echo "pmc pass $i ($G) rc=$?"
done
cd $R
python tools/pmc_summary.py $TAG /tmp/pmc_1 /tmp/pmc_2 gpurun_out/pmc_${TAG}_traffic.json
python tools/pmc_agg.py gpurun_out/pmc_${TAG}_counters.json /tmp/pmc_3 /tmp/pmc_4
