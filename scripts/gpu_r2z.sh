#!/bin/bash
# round-2 GPU pass Z: ln_gemv2 with 4 rows in flight per wave, greedy token-boundary kernel, split-K rule (>= 16 k-tiles per split): tests + cfg4 A/B
TAG=${1:-r2z}
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "decode or mmu or generate or splitk" 2>&1 | tail -3
for v in "4 1" "2 1" "4 0" "2 0" "4 1"; do
set -- $v
SHOWO_DECODE_LNR=$1 SHOWO_DECODE_TOKSEAM=$2 timeout 300 python bench.py --workload mmu --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/bench_mmu_${TAG}_$1_$2.log 2>&1
echo "lnr=$1 tokseam=$2"; grep -h '"metric"' gpurun_out/bench_mmu_${TAG}_$1_$2.log | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); c = d['config']; print(round(d['value'],1), 'tok/s', 'ttft', round(c['time_to_first_token_ms'],2), 'ms/token', round(c['ms_per_decoded_token'],4))
"
done
