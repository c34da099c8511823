#!/bin/bash
# round 5, pass C: batched decode with register-resident activations + co-scheduled fc2, A/B of both switches
mkdir -p gpurun_out/r5c
timeout 900 python -m pytest tests/test_precise_gpu.py tests/test_decode_batch_gpu.py -q > gpurun_out/r5c/tests.log 2>&1; echo "tests rc=$?"; tail -6 gpurun_out/r5c/tests.log
timeout 900 python -m pytest tests/test_modules_gpu.py -q -s -k "cfg4_mmu_vit" > gpurun_out/r5c/cfg4_tests.log 2>&1; echo "cfg4 rc=$?"; grep -E "4 sequences|passed|failed|^E  " gpurun_out/r5c/cfg4_tests.log | tail -5
for tag in default co0 reg0; do
  case $tag in default) envs="";; co0) envs="SHOWO_DECODE_BATCH_CO=0";; reg0) envs="SHOWO_DECODE_BATCH_REG=0";; esac
  env $envs timeout 600 python bench.py --workload mmu --steps 2 --warmup 1 > gpurun_out/r5c/mmu_$tag.json 2> gpurun_out/r5c/mmu_$tag.log; echo "mmu $tag rc=$?"
  python - $tag <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/r5c/mmu_{sys.argv[1]}.json").read().strip().splitlines()[-1])
b4, b1 = d["config"]["batch4"], d["config"]["batch1"]
print(sys.argv[1], "batch4 agg tok/s %.0f  step ms %.3f  hbm %.0f GB/s  | batch1 tok/s %.0f hbm %.0f" % (b4["aggregate_tokens_per_s"], b4["ms_per_step_of_4_tokens"], b4["hbm_GBps"], b1["tokens_per_s"], b1["hbm_GBps"]))
PY
done
