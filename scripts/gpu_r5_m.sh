#!/bin/bash
# round 5, pass M: resident attention kernel (parity, same-box A/B on the headline and training lines), decode GEMVs with wave-uniform bases
mkdir -p gpurun_out/r5m
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "attention or decode or qk_prep or kv_cache" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_modules_gpu.py -q -k "decode_layer or tiny_mmu or tiny_forward or tiny_t2i or full_size_t2i" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_train_gpu.py -q -k "tiny_training_step or attention" 2>&1 | tail -2
for tag in res off res2; do
  case $tag in res*) envs="A=1";; off) envs="SHOWO_ATTN_RES=0";; esac
  env $envs timeout 600 python bench.py --steps 6 --warmup 2 --no-train-leg --no-config-legs --no-cpu-baseline --no-accuracy-leg > gpurun_out/r5m/bench_$tag.json 2> gpurun_out/r5m/bench_$tag.log
  python - $tag <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/r5m/bench_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print("%-6s images/s %.2f  gemm frac %.3f  attn TF/s %.0f" % (sys.argv[1], d["value"], d["roofline"]["frac"], d["roofline"]["attention"]["achieved"]))
PY
done
for tag in res off; do
  case $tag in res*) envs="A=1";; off) envs="SHOWO_ATTN_RES=0";; esac
  env $envs timeout 600 python bench.py --workload train --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r5m/train_$tag.json 2> gpurun_out/r5m/train_$tag.log
  python - $tag <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/r5m/train_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print("%-6s train ms/step %.2f" % (sys.argv[1], d["ms_per_step"]))
PY
done
timeout 600 python bench.py --workload mmu --steps 2 --warmup 1 > gpurun_out/r5m/mmu.json 2> gpurun_out/r5m/mmu.log
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5m/mmu.json").read().strip().splitlines()[-1])
b4, b1 = d["config"]["batch4"], d["config"]["batch1"]
print("batch4 agg tok/s %.0f  step ms %.3f | batch1 tok/s %.0f  hbm %.0f GB/s  prefill->first ms %.2f" % (b4["aggregate_tokens_per_s"], b4["ms_per_step_of_4_tokens"], b1["tokens_per_s"], b1["hbm_GBps"], d["config"]["prefill_to_first_token_ms"]))
PY
