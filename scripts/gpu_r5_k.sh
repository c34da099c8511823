#!/bin/bash
# round 5, pass K: CLIP lazy low halves + clip kernel tests; attention launch-bounds A/B on the headline line (same box)
mkdir -p gpurun_out/r5k
timeout 900 python -m pytest tests/test_clip_gpu.py tests/test_kernels_gpu.py -q -k "clip or grad_clip" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_modules_gpu.py -q -k "cfg4_mmu_vit" 2>&1 | tail -2
for tag in default attn3 default2; do
  case $tag in default*) envs="A=1";; attn3) envs="SHOWO_ATTN_IMPL=3";; esac
  env $envs timeout 600 python bench.py --steps 6 --warmup 2 --no-train-leg --no-config-legs --no-cpu-baseline --no-accuracy-leg > gpurun_out/r5k/bench_$tag.json 2> gpurun_out/r5k/bench_$tag.log
  python - $tag <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/r5k/bench_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print("%-9s images/s %.2f  gemm frac %.3f  attn TF/s %.0f  vq conv TF/s %.0f" % (sys.argv[1], d["value"], d["roofline"]["frac"], d["roofline"]["attention"]["achieved"], d["roofline"]["vq_conv"]["achieved"]))
PY
done
