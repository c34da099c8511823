#!/bin/bash
# round 5, pass N: batched decode after the cleanup (bits), role-block counts beyond one block per CU
mkdir -p gpurun_out/r5n
timeout 600 python -m pytest tests/test_decode_batch_gpu.py -q 2>&1 | tail -1
timeout 600 python -m pytest tests/test_modules_gpu.py -q -k cfg4_mmu_vit 2>&1 | tail -1
for cb in 128 160 192 256; do
  SHOWO_DECODE_BATCH_CO_BLOCKS=$cb timeout 600 python bench.py --workload mmu --steps 2 --warmup 1 > gpurun_out/r5n/mmu_$cb.json 2> gpurun_out/r5n/mmu_$cb.log
  python - $cb <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/r5n/mmu_{sys.argv[1]}.json").read().strip().splitlines()[-1])
b4, b1 = d["config"]["batch4"], d["config"]["batch1"]
print("co_blocks %-4s batch4 agg tok/s %.0f  step ms %.3f | batch1 tok/s %.0f" % (sys.argv[1], b4["aggregate_tokens_per_s"], b4["ms_per_step_of_4_tokens"], b1["tokens_per_s"]))
PY
done
