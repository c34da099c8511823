#!/bin/bash
# round 5, pass I: fc2 role variants of the batched decode (bits first, then speed), kernel table of the best one
R=$(pwd)
mkdir -p gpurun_out/r5i
for r2 in 1 0; do
  SHOWO_DECODE_BATCH_ROWS2=$r2 timeout 600 python -m pytest tests/test_decode_batch_gpu.py -q 2>&1 | tail -1
  SHOWO_DECODE_BATCH_ROWS2=$r2 timeout 600 python -m pytest tests/test_modules_gpu.py -q -k cfg4_mmu_vit 2>&1 | tail -1
  SHOWO_DECODE_BATCH_ROWS2=$r2 timeout 600 python bench.py --workload mmu --steps 2 --warmup 1 > gpurun_out/r5i/mmu_rows$r2.json 2> gpurun_out/r5i/mmu_rows$r2.log
  python - rows$r2 <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/r5i/mmu_{sys.argv[1]}.json").read().strip().splitlines()[-1])
b4, b1 = d["config"]["batch4"], d["config"]["batch1"]
print("%-10s batch4 agg tok/s %.0f  step ms %.3f  hbm %.0f GB/s  | batch1 tok/s %.0f" % (sys.argv[1], b4["aggregate_tokens_per_s"], b4["ms_per_step_of_4_tokens"], b4["hbm_GBps"], b1["tokens_per_s"]))
PY
done
export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r5i/prof_mmu -o prof -- python $R/bench.py --workload mmu --steps 1 --warmup 1 > $R/gpurun_out/r5i/prof_mmu.log 2>&1
cd $R
find gpurun_out/r5i/prof_mmu -type f ! -name "*stats*" -size +2M -delete
f=$(find gpurun_out/r5i/prof_mmu -name "*kernel_stats.csv" | head -1); grep -E "gemvB|coB|y2B|seam_rows|ln_gemv2|decode_co_kernel|out_gemv2" $f | cut -c1-140
