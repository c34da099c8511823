#!/bin/bash
# round 5, pass B: remaining accuracy-mode kernel tests, the batched decode (tiny + full-size cfg4), the mmu bench line with batch 1 and batch 4
mkdir -p gpurun_out/r5b
timeout 900 python -m pytest tests/test_precise_gpu.py -q > gpurun_out/r5b/precise_tests.log 2>&1; echo "precise rc=$?"; tail -4 gpurun_out/r5b/precise_tests.log
timeout 600 python -m pytest tests/test_decode_batch_gpu.py -q > gpurun_out/r5b/batch_tests.log 2>&1; echo "batch rc=$?"; tail -6 gpurun_out/r5b/batch_tests.log
timeout 900 python -m pytest tests/test_modules_gpu.py -q -s -k "cfg4_mmu_vit or tiny_mmu or decode_layer" > gpurun_out/r5b/cfg4_tests.log 2>&1; echo "cfg4 rc=$?"; grep -E "parity\]|passed|failed|^E  " gpurun_out/r5b/cfg4_tests.log | tail -12
timeout 600 python bench.py --workload mmu --steps 2 --warmup 1 > gpurun_out/r5b/mmu_bench.json 2> gpurun_out/r5b/mmu_bench.log; echo "mmu bench rc=$?"; tail -3 gpurun_out/r5b/mmu_bench.log
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5b/mmu_bench.json").read().strip().splitlines()[-1])
print("value", d["value"], d["unit"]); print(json.dumps(d["config"]["batch4"], indent=1)); print(json.dumps(d["config"]["batch1"], indent=1))
PY
