#!/bin/bash
# round-4 pass K (VERDICT r3 #3c): 2-product split (hi*hi + hi*lo) in the VQGAN conv -- errors / id agreement on the reference fixtures and speed
for p in 3 2; do
  echo "== SHOWO_CONV_PRODUCTS=$p"
  SHOWO_CONV_PRODUCTS=$p timeout 200 python -m pytest tests/test_modules_gpu.py -m gpu -q -rP -k "magvit" 2>&1 | grep -E "^\[parity\] magvit.*(precision=1|256x256)|passed|failed" | cut -c1-200
  SHOWO_CONV_PRODUCTS=$p timeout 120 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-train-leg --no-accuracy-leg --no-config-legs 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('t2i PRODUCTS=$p', round(d['value'], 2), 'images/s  vq_conv', round(d['roofline']['vq_conv']['achieved'], 1), 'TF/s')
"
done
