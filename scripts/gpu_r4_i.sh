#!/bin/bash
# round-4 pass I: coalesced LFQ pack for NHWC latents: bit-exact tests + the vq bandwidth line
timeout 300 python -m pytest tests -m gpu -q -x -k "lfq or magvit or quantizer" 2>&1 | grep -E "passed|failed|error|^E  " | tail -4
timeout 200 python bench.py --workload vq --steps 5 --warmup 1 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('vq', round(d['value'], 1), d['unit']); print({k: round(v['GBps']) for k, v in d['config']['kernels'].items()})
"
