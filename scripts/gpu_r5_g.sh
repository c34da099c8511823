#!/bin/bash
# round 5, pass G: the split attention after the conversion fix (diagnostic + tests), bf16 attention tests, headline line
mkdir -p gpurun_out/r5g
timeout 300 python tools/diag_split_attn.py 2>&1 | grep -E "random|causal|flags|full|bf16 kernel|interval err" | head -12
timeout 900 python -m pytest tests/test_precise_gpu.py tests/test_decode_batch_gpu.py -q 2>&1 | tail -4
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "attention or qk_prep or decode" 2>&1 | tail -3
timeout 600 python bench.py --steps 5 --warmup 2 --no-train-leg --no-config-legs --no-cpu-baseline > gpurun_out/r5g/bench.json 2> gpurun_out/r5g/bench.log; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5g/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "gemm frac", d["roofline"]["frac"], "attn TF", d["roofline"]["attention"]["achieved"])
a = d["accuracy_mode"]; print("accuracy", a["images_per_s"], a["roofline"]["frac"], a["roofline"]["attention"])
PY
