#!/bin/bash
# round 5, pass J: CU-mask / grid-budget / cooperative-ownership tests, the clip kernel, batched decode after the role rework,
# the exchange-contention experiment with grids sized to 256 - r CUs
mkdir -p gpurun_out/r5j
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "cu_masked or two_streams or grad_clip or splitk or gemm_both or kcat" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_train_gpu.py -q -k "clip or exchange or adamw or tiny_training" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_decode_batch_gpu.py tests/test_precise_gpu.py -q 2>&1 | tail -2
timeout 600 python bench.py --workload mmu --steps 2 --warmup 1 > gpurun_out/r5j/mmu.json 2> gpurun_out/r5j/mmu.log
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5j/mmu.json").read().strip().splitlines()[-1])
b4, b1 = d["config"]["batch4"], d["config"]["batch1"]
print("batch4 agg tok/s %.0f  step ms %.3f  hbm %.0f GB/s  | batch1 tok/s %.0f" % (b4["aggregate_tokens_per_s"], b4["ms_per_step_of_4_tokens"], b4["hbm_GBps"], b1["tokens_per_s"]))
PY
timeout 900 python tools/exchange_contention.py > gpurun_out/r5j/exchange_contention.txt 2> gpurun_out/r5j/exchange_contention.log; echo "contention rc=$?"; tail -12 gpurun_out/r5j/exchange_contention.txt
