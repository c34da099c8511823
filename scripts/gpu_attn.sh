#!/bin/bash
# attention forward at the t2i shape: micro-benchmark (tools/attn_bench.py) over variants / shapes + one SQ PMC pass.  usage: gpu_attn.sh <tag> [impl list for showo_attn_set_impl, e.g. 0,3]
TAG=${1:-attn}; VAR=${2:-0}
R=$(pwd); mkdir -p gpurun_out; export TMPDIR=/tmp
{
for op in 0 1; do
  for lq in 258 256 288; do
    echo "== op $op Lq $lq"; timeout 300 python tools/attn_bench.py --variants $VAR --op $op --Lq $lq --Lk $((lq + 129))
  done
done
} > gpurun_out/${TAG}_attn_bench.txt 2>&1
cat gpurun_out/${TAG}_attn_bench.txt
cd /tmp; rm -rf /tmp/pmc_attn
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d /tmp/pmc_attn -o pmc -- python $R/tools/attn_bench.py --variants $VAR --once > $R/gpurun_out/${TAG}_attn_pmc.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU --output-format csv -d /tmp/pmc_attn2 -o pmc -- python $R/tools/attn_bench.py --variants $VAR --once >> $R/gpurun_out/${TAG}_attn_pmc.log 2>&1
cd $R
python tools/pmc_agg.py gpurun_out/${TAG}_attn_pmc.json /tmp/pmc_attn /tmp/pmc_attn2 | grep -i attn
