#!/bin/bash
# round 5, pass E: knob sweep of the batched decode (one box, same build) + the split-attention dense diagnostic after the aliasing fix
mkdir -p gpurun_out/r5e
timeout 300 python tools/diag_split_attn.py 2>&1 | tail -6
run() {
  tag=$1; shift
  env "$@" timeout 600 python bench.py --workload mmu --steps 2 --warmup 1 > gpurun_out/r5e/mmu_$tag.json 2> gpurun_out/r5e/mmu_$tag.log
  python - $tag <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/r5e/mmu_{sys.argv[1]}.json").read().strip().splitlines()[-1])
b4, b1 = d["config"]["batch4"], d["config"]["batch1"]
print("%-14s batch4 agg tok/s %.0f  step ms %.3f  hbm %.0f GB/s  | batch1 tok/s %.0f" % (sys.argv[1], b4["aggregate_tokens_per_s"], b4["ms_per_step_of_4_tokens"], b4["hbm_GBps"], b1["tokens_per_s"]))
PY
}
run default A=1
run co64 SHOWO_DECODE_BATCH_CO_BLOCKS=64
run co128 SHOWO_DECODE_BATCH_CO_BLOCKS=128
run r4 SHOWO_DECODE_BATCH_R=4
run ln256 SHOWO_DECODE_BATCH_LNBLOCKS=256
run ln1024 SHOWO_DECODE_BATCH_LNBLOCKS=1024
run r4co128 SHOWO_DECODE_BATCH_R=4 SHOWO_DECODE_BATCH_CO_BLOCKS=128
