#!/bin/bash
# round 5, pass H: full GPU suite on the compiler-visible bf16 conversion + grouped reductions, mmu bench (default and 128 fc2 blocks)
mkdir -p gpurun_out/r5h
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r5h/gpu_tests.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/r5h/gpu_tests.log | tail -3
grep -E "^(FAILED|ERROR)|^E  " gpurun_out/r5h/gpu_tests.log | head -10
for tag in default co128; do
  case $tag in default) envs="A=1";; co128) envs="SHOWO_DECODE_BATCH_CO_BLOCKS=128";; esac
  env $envs timeout 600 python bench.py --workload mmu --steps 2 --warmup 1 > gpurun_out/r5h/mmu_$tag.json 2> gpurun_out/r5h/mmu_$tag.log
  python - $tag <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/r5h/mmu_{sys.argv[1]}.json").read().strip().splitlines()[-1])
b4, b1 = d["config"]["batch4"], d["config"]["batch1"]
print("%-10s batch4 agg tok/s %.0f  step ms %.3f  hbm %.0f GB/s  | batch1 tok/s %.0f" % (sys.argv[1], b4["aggregate_tokens_per_s"], b4["ms_per_step_of_4_tokens"], b4["hbm_GBps"], b1["tokens_per_s"]))
PY
done
