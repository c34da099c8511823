#!/bin/bash
# round 5, pass A: accuracy mode on the production kernels -- kernel parity, the full-size fixtures, and the bench's accuracy leg
mkdir -p gpurun_out/r5a
timeout 600 python -m pytest tests/test_precise_gpu.py -x -q > gpurun_out/r5a/precise_tests.log 2>&1; echo "precise rc=$?"; tail -4 gpurun_out/r5a/precise_tests.log
timeout 900 python -m pytest tests/test_modules_gpu.py -q -s -k "full_size_logits_vs_reference_subset or cfg3_inpainting or cfg4_mmu_vit or t2i_generate_is_reproducible or tiny_accuracy" > gpurun_out/r5a/fullsize_tests.log 2>&1; echo "fullsize rc=$?"; grep -E "parity\]|passed|failed|^E  " gpurun_out/r5a/fullsize_tests.log | tail -40
SHOWO_GEMM_TUNE_LOG=1 timeout 600 python bench.py --steps 5 --warmup 2 --no-train-leg --no-config-legs --no-cpu-baseline > gpurun_out/r5a/bench.json 2> gpurun_out/r5a/bench.log; echo "bench rc=$?"
grep -E "accuracy-mode|timed|tune\]" gpurun_out/r5a/bench.log | tail -30
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5a/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "roofline", d["roofline"]["frac"])
print("accuracy", json.dumps(d["accuracy_mode"], indent=1)[:3000])
PY
