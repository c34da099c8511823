#!/bin/bash
# round-3 GPU pass C: attention block order (XCD-aware / 9-wave blocks) A/B, rewritten qkln_rope_bwd, training step time
TAG=${1:-r3c}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_train_gpu.py -m gpu -x -q -k "attn or attention or qk_layernorm or tiny_training or small_training" 2>&1 | tail -4
SHOWO_ATTN_WPB9=1 timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_train_gpu.py -m gpu -x -q -k "attn or attention" 2>&1 | tail -3
for cfg in "SHOWO_ATTN_XCD=0" "SHOWO_ATTN_XCD=1" "SHOWO_ATTN_WPB9=1" "SHOWO_ATTN_XCD=0" "SHOWO_ATTN_XCD=1" "SHOWO_ATTN_WPB9=1"; do
  env $cfg timeout 300 python bench.py --steps 6 --warmup 2 --no-train-leg --no-cpu-baseline > gpurun_out/${TAG}_attn.log 2>&1
  echo "$cfg"; grep -h '"metric"' gpurun_out/${TAG}_attn.log | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d['roofline']; print(round(d['value'], 2), 'img/s  gemm', round(r['achieved']), 'TF/s  attention', round(r['attention']['achieved'], 1), 'TF/s')
"
done
timeout 400 python bench.py --workload train --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_train.log 2>&1; grep -h '"metric"' gpurun_out/${TAG}_train.log | tail -1 > gpurun_out/${TAG}_train_bench.json; cut -c 1-330 gpurun_out/${TAG}_train_bench.json
